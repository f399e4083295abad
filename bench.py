#!/usr/bin/env python3
"""bench.py -- frames/sec of enhance (FastDVDnet) + detect (RetinaNet R-50+FPN, decode, NMS, D2H) on synthetic
1280x720 dark clips, one process per GPU, no data-path collective (clips shard across ranks: weak scaling).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over `--frames-per-step` (default 16: THROUGHPUT mode) consecutive frames of a clip
that is already resident in HBM (preprocessed: Brighten -> Normalize -> Pad, as the reference's test pipeline does
before the model; the reference's benchmark.py excludes data loading too). Timed: FastDVDnet reading its window
straight out of the resident clip (B+2 temp1 DenBlocks + B temp2 DenBlocks for the B frames of the step: every
distinct temp1 result is computed once per step, nothing is carried between steps; B=1 is the reference's 3+1 per
frame), ResNet-50, FPN, RetinaHead, decode/top-k, NMS and the D2H copy of the detections. The JSON line also carries
the reference's own per-frame protocol (benchmark.py:83-109: batch 1, >= 200 frames) as `latency_mode` (4 DenBlocks
per frame) and `streaming_mode` (consecutive frames, 2 DenBlocks per frame). Rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts the N ranks itself
(`python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>`, child processes, before any GPU
call here), forwards their output and exits with their return code. Under torchrun, WORLD_SIZE must equal --gpus.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = 'frames/sec (enhance+detect) on 1280x720 dark video'
NUM_CLASSES = 80


def retinanet_cfg():
    """mmdetection/configs/_base_/models/retinanet_r50_fpn.py:2-60 (pretrained=None: no network)."""
    return dict(
        type='RetinaNet', pretrained=None,
        backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'),
        neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                  add_extra_convs='on_input', num_outs=5),
        bbox_head=dict(type='RetinaHead', num_classes=NUM_CLASSES, in_channels=256, stacked_convs=4, feat_channels=256,
                       anchor_generator=dict(type='AnchorGenerator', octave_base_scale=4, scales_per_octave=3,
                                             ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128]),
                       bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0],
                                       target_stds=[1.0, 1.0, 1.0, 1.0]),
                       loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                       loss_bbox=dict(type='L1Loss', loss_weight=1.0)),
        train_cfg=None,
        test_cfg=dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5),
                      max_per_img=100))


def build_models(seed):
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    from lowlightenvironmentvideoobjectdetection_amd.synthetic import seeded_weights
    det = seeded_weights(registry.build_detector(retinanet_cfg()), seed=seed)
    cleaner = seeded_weights(registry.build_cleaner(dict(type='FastDVDnet')), seed=seed + 1)
    return det, cleaner


def cpu_baseline(det, cleaner, clip_cpu, img_shape):
    """The oracle (CPU restatement of the reference path) timed on this host: ONE frame of the same workload
    (5-frame window -> FastDVDnet -> RetinaNet -> decode -> NMS), all cores."""
    from oracle.cleaner import fastdvdnet_forward
    from oracle.detector import retinanet_simple_test
    # the GPU box gives one GPU job a 16-core share of the host; use what the affinity mask allows, at most 16
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    sd_c = {k: v.detach().float().cpu() for k, v in cleaner.state_dict().items()}
    sd_d = {k: v.detach().float().cpu() for k, v in det.state_dict().items()}
    win = clip_cpu[0:5, :, :, :3].float().permute(0, 3, 1, 2).reshape(1, 15, clip_cpu.shape[1], clip_cpu.shape[2]).contiguous()
    meta = dict(img_shape=img_shape, scale_factor=[1.0, 1.0, 1.0, 1.0])
    with torch.no_grad():
        t0 = time.perf_counter()
        den = fastdvdnet_forward(sd_c, win)
        t1 = time.perf_counter()
        retinanet_simple_test(sd_d, den, meta, num_classes=NUM_CLASSES)
        t2 = time.perf_counter()
    return dict(value=1.0 / (t2 - t0), unit='frames/s', cores=cores, kind='port',
                sample=f'1 frame 1280x720 (padded {clip_cpu.shape[1]}x{clip_cpu.shape[2]}): FastDVDnet {t1 - t0:.2f} s + '
                       f'RetinaNet R-50+FPN/decode/NMS {t2 - t1:.2f} s, torch fp32, {cores} threads')


def _profiled_traffic(frames_per_step, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/collect_profile.sh: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate --pmc passes), if they were taken at this
    frames_per_step; None otherwise."""
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        t = json.load(open(path))
        if int(t.get('frames_per_step', -1)) == int(frames_per_step):
            return t['per_kernel'][kernel]['hbm_bytes_per_launch']
    except Exception:
        pass
    return None


def train_bench(args, world, rank, dev):
    """Config #3 of BASELINE.json: key frames of clips, one (or --train-batch) 1280x720 image per GPU, k = 10 ground-truth
    boxes per image, RetinaNet R-50+FPN forward + focal/L1 loss + backward (stem and layer1 frozen, norm_eval) +
    bucketed gradient all-reduce over the ranks + SGD(momentum 0.9, weight decay 1e-4). Synthetic data, seeded weights."""
    import numpy as np
    import torch.distributed as dist
    from lowlightenvironmentvideoobjectdetection_amd import data, train_engine
    cfg = retinanet_cfg()
    cfg['train_cfg'] = dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1),
                            allowed_border=-1, pos_weight=-1, debug=False)
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    from lowlightenvironmentvideoobjectdetection_amd.synthetic import seeded_weights
    det = seeded_weights(registry.build_detector(cfg), seed=0).to(dev)
    det.train()
    nb = args.train_batch
    clean = data.synth_clean_clip(nb, args.height, args.width, seed=100 + rank)
    clip, img_shape, pad_shape = data.preprocess_clip(data.darken_and_add_noise(clean, seed=200 + rank), dev)
    img = clip[2:2 + nb, :, :, :3].permute(0, 3, 1, 2).float().contiguous()          # (nb,3,736,1280) normalised
    hp, wp = img.shape[2], img.shape[3]
    rs = np.random.RandomState(rank)                                                  # _demo_mm_inputs-style boxes (test_forward.py:326-401)
    gtb, gtl = [], []
    for _ in range(nb):
        cx, cy = rs.rand(10) * args.width, rs.rand(10) * args.height
        bw, bh = rs.rand(10) * args.width * 0.4 + 16, rs.rand(10) * args.height * 0.4 + 16
        bx = np.stack([np.clip(cx - bw / 2, 0, args.width), np.clip(cy - bh / 2, 0, args.height),
                       np.clip(cx + bw / 2, 0, args.width), np.clip(cy + bh / 2, 0, args.height)], 1).astype(np.float32)
        gtb.append(torch.from_numpy(bx).to(dev))
        gtl.append(torch.from_numpy(rs.randint(0, NUM_CLASSES, 10)).long().to(dev))
    metas = [dict(img_shape=img_shape, pad_shape=pad_shape, scale_factor=1.0)] * nb
    ts = train_engine.TrainStep(det, nb, hp, wp, dev, lr=0.01 / 16 * nb * world,      # linear scaling from 16 images at lr 0.01
                                bucket_mb=float(os.environ.get('LLV_BUCKET_MB', '18')))
    for _ in range(args.warmup):
        out = ts.step_fused(img, metas, gtb, gtl)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = ts.step_fused(img, metas, gtb, gtl)
    t_enq = time.perf_counter() - t0             # host time to enqueue the steps (no synchronisation inside a step)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        g = ts.graph
        fwd = sum(st[3] for st in g.plan.steps)
        trainable_fwd = sum(2.0 * sum(op['batch'] * sg[2] * sg[3] for sg in op['segs']) * op['pc'].true_macs_per_pixel
                            for op in g.tape if not op['frozen'])
        dgrad_fl = sum(2.0 * sum(op['batch'] * sg[2] * sg[3] for sg in op['segs']) * op['pc'].true_macs_per_pixel
                       for op in g.tape if op['need_dx'])
        flops = fwd + trainable_fwd + dgrad_fl
        print(json.dumps({
            'metric': 'training images/sec (RetinaNet R-50+FPN forward + loss + backward + gradient all-reduce + SGD)',
            'value': round(args.steps * nb * world / dt, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': f'config #3: {nb} image(s) {args.width}x{args.height} (padded {wp}x{hp}) per GPU, 10 gt boxes each, '
                                   f'{NUM_CLASSES} classes, frozen stem+layer1, norm_eval, SGD momentum; loss scale 512',
                       'parallelism': f'data parallel x{world}: bucketed all-reduce of {ts.flat_g.numel() * 4 / 1e6:.0f} MB fp32 gradients '
                                      f'in {len(ts.buckets)} buckets, overlapped with backward',
                       'algorithmic_gflop_per_image': round(flops / nb / 1e9, 1), 'last_loss': out['log_vars']['loss']},
            'achieved_tflops': round(flops * args.steps / dt / 1e12, 2),
            'host_enqueue_ms_per_step': round(t_enq / args.steps * 1e3, 4)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def yolo_bench(args, world, rank, dev):
    """Config #5 of BASELINE.json: YOLO-style detector (the reference's YOLOv3-D53 stand-in) on 1920x1080
    frames (padded 1920x1088), fp16, --frames-per-step frames per step: convs + decode as one HIP graph, then the per-level
    top-k / NMS. Seeded random weights (no checkpoints offline), synthetic frames."""
    import numpy as np
    import torch.distributed as dist
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import data, ops, registry
    from lowlightenvironmentvideoobjectdetection_amd.synthetic import seeded_weights
    cfg = dict(type='YOLOV3', pretrained=None, backbone=dict(type='Darknet', depth=53, out_indices=(3, 4, 5)),
               neck=dict(type='YOLOV3Neck', num_scales=3, in_channels=[1024, 512, 256], out_channels=[512, 256, 128]),
               bbox_head=dict(type='YOLOV3Head', num_classes=NUM_CLASSES, in_channels=[512, 256, 128], out_channels=[1024, 512, 256]),
               test_cfg=dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, conf_thr=0.005, nms=dict(type='nms', iou_threshold=0.45),
                             max_per_img=100))
    det = seeded_weights(registry.build_detector(cfg), seed=0)
    with torch.no_grad():
        for m in det.bbox_head.convs_pred:
            m.weight.mul_(0.2)
    det = det.to(dev).eval()
    B, H, W = args.frames_per_step, 1080, 1920
    clean = data.synth_clean_clip(B, H, W, seed=100 + rank)
    clip, img_shape, pad_shape = data.preprocess_clip(data.darken_and_add_noise(clean, seed=200 + rank), dev)
    img = clip[2:2 + B, :, :, :3].permute(0, 3, 1, 2).float().contiguous()
    plan = det._get_plan(img)
    ops.pack_input(img, 8, out=plan.x_in)
    plan.plan.capture()
    def step():
        plan.plan.replay()
        return plan.results()
    for _ in range(args.warmup):
        res = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({
            'metric': 'frames/sec (YOLOv3-D53 stand-in) on 1920x1080 video', 'value': round(args.steps * B * world / dt, 3), 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': f'config #5: YOLOv3 Darknet-53 + neck + head + decode + NMS, {W}x{H} frames (padded {img.shape[3]}x{img.shape[2]}), '
                                   f'{NUM_CLASSES} classes', 'frames_per_step': B, 'hip_graph': True,
                       'algorithmic_gflop_per_frame': round(plan.plan.flops / B / 1e9, 1), 'detections_last_step': int(sum(len(d) for d, _ in res))},
            'achieved_tflops': round(plan.plan.flops * args.steps / dt / 1e12, 2)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def insert_bench(args, world, rank, dev):
    """the authors' in-backbone denoising (InsertResNet-50, DenoisingAggregator with 8 dense layers,
    3 embedding convs and the DCN-based temporal attention fusion after every stage, as configs/vid/llvod/
    llvod_insert_plugins_l34_i1234_vid_a7s3.py:14-31 sets it, stride-2 stages instead of the dilated last one) on a clip of
    --frames-per-step 1280x720 frames (default 3: key frame + 2 references). One HIP graph per step. The JSON line carries the
    HBM roofline of the deformable-conv column kernel (llv_modulated_deform_im2col), timed per launch with HIP events."""
    import torch.distributed as dist
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import data, engine, ops, registry
    from lowlightenvironmentvideoobjectdetection_amd.synthetic import seeded_weights
    B = args.frames_per_step if args.frames_per_step != 16 else 3
    m = seeded_weights(registry.build_backbone(dict(
        type='InsertResNet', depth=50, num_stages=4, out_indices=(2, 3, 3), insert_tag=[True, True, True, True],
        insert_cfg=dict(type='DenoisingAggregator', num_blocks=1, num_layers=8, emb_nums=3, with_rdb=True, with_taf=True))), seed=0)
    with torch.no_grad():
        for name, mod in m.named_modules():
            if name.endswith('dcn_pack'):
                mod.conv_offset.weight.mul_(0.25)          # offsets of about a pixel
    clean = data.synth_clean_clip(B, args.height, args.width, seed=100 + rank)
    clip, img_shape, pad_shape = data.preprocess_clip_device(data.darken_and_add_noise(clean, seed=200 + rank).to(torch.uint8).to(dev))
    hp, wp = pad_shape[:2]
    b = engine.Builder(dev)
    x_in = b.buf(B, hp, wp, 8)
    sd = engine._cpu_sd(m.state_dict())

    def after_stage(i, nb, t):
        return engine.build_denoising_aggregator(b, sd, t, B, p=f'layer{i + 1}.{nb}.', tag=f'layer{i + 1}.plugin',
                                                 **m._plugins[i].plan_kwargs())
    engine.build_resnet(b, sd, x_in, after_stage=after_stage, **m.plan_kwargs())
    plan = b.plan
    x_in.copy_(clip[2:2 + B])
    plan.capture()
    for _ in range(args.warmup):
        plan.replay()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.replay()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        timed = plan.run_timed(iters=5)
        dcn = [(n, ms, nb) for (n, ms, fl, nb) in timed if n.endswith('dcn_im2col')]
        dms, dnb = sum(x[1] for x in dcn), sum(x[2] for x in dcn)
        plug_ms = sum(ms for (n, ms, _, _) in timed if '.plugin' in n)
        conv_ms = sum(ms for (n, ms, fl, _) in timed if fl > 0)
        conv_fl = sum(fl for (n, ms, fl, _) in timed if fl > 0)
        print(json.dumps({
            'metric': 'frames/sec (InsertResNet-50 backbone with DenoisingAggregator plugins) on 1280x720 clips',
            'value': round(args.steps * B * world / dt, 3), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': f'SURVEY 8f.3: InsertResNet-50 (plugin after each of the 4 stages: 8 dense layers, 3 embedding convs, '
                                   f'DCN temporal attention fusion over the {B} frames of the clip), {args.width}x{args.height} frames '
                                   f'(padded {wp}x{hp})', 'frames_per_step': B, 'hip_graph': True,
                       'algorithmic_gflop_per_step': round(plan.flops / 1e9, 1), 'plugin_ms_per_step': round(plug_ms, 3),
                       'launches_per_step': len(plan.steps)},
            'roofline': {'bound': 'hbm', 'kernel': 'mdcn_im2col_kernel', 'launches_per_step': len(dcn),
                         'achieved': round(dnb / (dms * 1e-3) / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': round(dnb / (dms * 1e-3) / 1e9 / 8000.0, 4), 'traffic': None,
                         'algorithmic_bytes_per_step': round(dnb), 'ms_per_step': round(dms, 4),
                         'all_conv_kernels': {'achieved': round(conv_fl / (conv_ms * 1e-3) / 1e12, 1), 'unit': 'TFLOP/s',
                                              'ms_per_step': round(conv_ms, 3)}}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def selsa_bench(args, world, rank, dev):
    """the authors' two-stage SELSA video detector at test time (FasterRCNN with a DC5 ResNet-50
    (strides (1,2,2,1), dilations (1,1,1,2)), ChannelMapper 2048 -> 512, RPNHead (12 anchors, nms_pre 6000, 300 proposals),
    SelsaRoIHead with two 1024-wide shared fcs and SelsaAggregators, 30 classes: detector of mmtracking/configs/vid/llvod/
    llvod_insert_plugins_l34_i1234_vid_a7s3.py:12-86,129-139 without the backbone plugins and with SingleRoIExtractor), one 1280x720
    frame per step, 2 reference frames given with frame 0 and kept in the memo. Module-level calls (NCHW at every module boundary, no
    launch plan yet): a first number for this row, not a tuned path."""
    import torch.distributed as dist
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from lowlightenvironmentvideoobjectdetection_amd import data
    from lowlightenvironmentvideoobjectdetection_amd.synthetic import seeded_weights
    NC = 30
    cfg = dict(type='SELSA', pretrains=None, detector=dict(
        type='FasterRCNN', pretrained=None,
        backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(3,), strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2), frozen_stages=-1,
                      norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'),
        neck=dict(type='ChannelMapper', in_channels=[2048], out_channels=512, kernel_size=3),
        rpn_head=dict(type='RPNHead', in_channels=512, feat_channels=512,
                      anchor_generator=dict(type='AnchorGenerator', scales=[4, 8, 16, 32], ratios=[0.5, 1.0, 2.0], strides=[16]),
                      bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]),
                      loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                      loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0)),
        roi_head=dict(type='SelsaRoIHead',
                      bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=2),
                                              out_channels=512, featmap_strides=[16]),
                      bbox_head=dict(type='SelsaBBoxHead', num_shared_fcs=2, aggregator=dict(type='SelsaAggregator', in_channels=1024, num_attention_blocks=16),
                                     in_channels=512, fc_out_channels=1024, roi_feat_size=7, num_classes=NC,
                                     bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=[0.2, 0.2, 0.2, 0.2]),
                                     reg_class_agnostic=False)),
        test_cfg=dict(rpn=dict(nms_across_levels=False, nms_pre=6000, nms_post=300, max_num=300, nms_thr=0.7, min_bbox_size=0),
                      rcnn=dict(score_thr=0.0001, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100))))
    model = M.build_model(cfg)
    seeded_weights(model.detector.backbone, seed=0)
    seeded_weights(model.detector.neck, seed=1)
    seeded_weights(model.detector.rpn_head, seed=2)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        model.detector.rpn_head.rpn_cls.bias.add_(-2.0)
        model.detector.rpn_head.rpn_reg.weight.mul_(0.3)
        for p in model.detector.roi_head.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * ((1.0 / p.shape[1]) ** 0.5 if p.dim() == 2 else 0.05))
    model = model.to(dev).eval()
    T = 8
    clean = data.synth_clean_clip(T, args.height, args.width, seed=100 + rank)
    clip, img_shape, pad_shape = data.preprocess_clip_device(data.darken_and_add_noise(clean, seed=200 + rank).to(torch.uint8).to(dev))
    frames = clip[2:2 + T, :, :, :3].permute(0, 3, 1, 2).float().contiguous()
    meta0 = dict(img_shape=img_shape, pad_shape=pad_shape, ori_shape=img_shape, scale_factor=np.ones(4, np.float32), flip=False)

    def step(i, first=False):
        meta = dict(meta0, frame_id=0 if first else 1 + i, num_left_ref_imgs=-1, frame_stride=-1)
        kw = dict(ref_img=[frames[:2].unsqueeze(0)], ref_img_metas=[[[meta0, meta0]]]) if first else {}
        with torch.no_grad():
            return model(return_loss=False, img=[frames[2 + i % (T - 2):3 + i % (T - 2)]], img_metas=[[meta]], **kw)['bbox_results']
    res = step(0, first=True)
    for i in range(args.warmup):
        res = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({
            'metric': 'frames/sec (SELSA two-stage video detector, test time) on 1280x720 video', 'value': round(args.steps * world / dt, 3),
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': f'SURVEY 8f.4: SELSA (FasterRCNN R-50 DC5 + ChannelMapper + RPN 300 proposals + SelsaRoIHead, {NC} classes), '
                                   f'{args.width}x{args.height} frames, 2 reference frames in the memo; frame 0 of the video through the module-level path, every key frame '
                                   f'after it = one captured launch list (selsa_plan.SelsaKeyPlan)',
                       'frames_per_step': 1, 'hip_graph': bool(model.use_key_plan), 'detections_last_step': int(sum(len(r) for r in res))}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def vid2s_bench(args, world, rank, dev, kind):
    """The reference's OWN video configs at test time: `configs/vid/dff/dff_faster_rcnn_r50_dc5_1x_imagenetvid.py` (kind 'dff2s': key frame
    every 10 frames, the others warp the key features with FlowNetSimple) and `configs/vid/fgfa/fgfa_faster_rcnn_r50_dc5_1x_imagenetvid.py`
    (kind 'fgfa2s': 30 reference frames around every frame, test sampler `test_with_fix_stride`, stride 1, EmbedAggregator over 512 channels)
    around `_base_/models/faster_rcnn_r50_dc5.py` (ResNet-50 DC5, ChannelMapper 2048 -> 512, RPNHead with 12 anchors and 300 proposals,
    StandardRoIHead + Shared2FCBBoxHead, 30 classes), 1280x720 frames, one frame per step, fed as the reference's test dataset feeds them
    (apis.ClipTestFeed). Module-level calls (no launch plan over the whole step yet): a first number for the two-stage branch."""
    import torch.distributed as dist
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from lowlightenvironmentvideoobjectdetection_amd import apis, data
    from lowlightenvironmentvideoobjectdetection_amd.synthetic import seeded_weights
    NC = 30
    det = dict(
        type='FasterRCNN', pretrained=None,
        backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(3,), strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2), frozen_stages=-1,
                      norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'),
        neck=dict(type='ChannelMapper', in_channels=[2048], out_channels=512, kernel_size=3),
        rpn_head=dict(type='RPNHead', in_channels=512, feat_channels=512,
                      anchor_generator=dict(type='AnchorGenerator', scales=[4, 8, 16, 32], ratios=[0.5, 1.0, 2.0], strides=[16]),
                      bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0])),
        roi_head=dict(type='StandardRoIHead',
                      bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=2),
                                              out_channels=512, featmap_strides=[16]),
                      bbox_head=dict(type='Shared2FCBBoxHead', in_channels=512, fc_out_channels=1024, roi_feat_size=7, num_classes=NC,
                                     bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=[0.2, 0.2, 0.2, 0.2]),
                                     reg_class_agnostic=False)),
        test_cfg=dict(rpn=dict(nms_across_levels=False, nms_pre=6000, nms_post=300, max_num=300, nms_thr=0.7, min_bbox_size=0),
                      rcnn=dict(score_thr=0.0001, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100)))
    flow = dict(type='FlowNetSimple', img_scale_factor=0.5)
    if kind == 'dff2s':
        cfg = dict(type='DFF', detector=det, motion=flow, test_cfg=dict(key_frame_interval=10))
        sampler = None
    else:
        cfg = dict(type='FGFA', detector=det, motion=flow, aggregator=dict(type='EmbedAggregator', num_convs=1, channels=512, kernel_size=3))
        sampler = dict(num_ref_imgs=30, frame_range=[-15, 15], stride=1, method='test_with_fix_stride')
    model = M.build_model(cfg)
    seeded_weights(model, seed=0)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        model.detector.rpn_head.rpn_cls.bias.add_(-2.0)
        model.detector.rpn_head.rpn_reg.weight.mul_(0.3)
        for p in model.detector.roi_head.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * ((1.0 / p.shape[1]) ** 0.5 if p.dim() == 2 else 0.05))
    model = model.to(dev).eval()
    assert model.two_stage
    warm = max(args.warmup, 10)                       # a whole key-frame period of DFF; FGFA's frame 0 builds the 31-frame window
    steps = (args.steps + 9) // 10 * 10
    T = warm + steps
    clean = data.synth_clean_clip(min(T + 16, 48), args.height, args.width, seed=100 + rank)
    clip, img_shape, pad_shape = data.preprocess_clip(data.darken_and_add_noise(clean, seed=200 + rank), dev, size_divisor=128)
    base = clip[:, :, :, :3].permute(0, 3, 1, 2).float().contiguous()
    frames = base[torch.arange(T + 16) % base.shape[0]]
    meta = dict(img_shape=img_shape, pad_shape=pad_shape, ori_shape=img_shape, scale_factor=[1.0, 1.0, 1.0, 1.0], flip=False,
                img_norm_cfg=dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True))
    feed = list(apis.make_feed(model, frames, meta, sampler, 0, T))
    apis.prepare_segment(model, apis.make_feed(model, frames, meta, sampler, 0, T))

    def step(i):
        with torch.no_grad():
            return model(return_loss=False, rescale=True, **feed[i][1])

    for i in range(warm):
        out = step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warm, T):
        out = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        what = ('DFF: key frame every 10 frames, FlowNetSimple + feature warp on the others' if kind == 'dff2s' else
                'FGFA: 30 reference frames per frame (fixed stride 1), flow + warp + EmbedAggregator over 512 channels')
        print(json.dumps({
            'metric': f'frames/sec ({what}; Faster R-CNN R-50 DC5 + StandardRoIHead) on 1280x720 video',
            'value': round(steps * world / dt, 3), 'unit': 'frames/s', 'n_gpus': world, 'steps': steps, 'warmup': warm,
            'ms_per_step': round(dt / steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': f"the reference's configs/vid/{'dff' if kind == 'dff2s' else 'fgfa'}/*_faster_rcnn_r50_dc5_1x_imagenetvid.py at "
                                   f'test time, 1 frame/step, frames padded to {pad_shape[1]}x{pad_shape[0]}, module-level eager launches',
                       'detections_last_step': int(sum(len(a) for a in out['bbox_results']))}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def fgfa_bench(args, world, rank, dev):
    """Config #4 of BASELINE.json: flow-guided feature aggregation over a 10-reference window (mmtracking FGFA with
    EmbedAggregator on the 256-channel FPN levels, RetinaNet R-50+FPN detector, FlowNetSimple), 1280x720 frames, one frame
    per step: backbone + FPN, 10 flow fields, 50 feature warps, embed convs over 11 x 5 levels, aggregation, head, decode,
    NMS. Eager launches (no graph). Seeded weights, synthetic clip."""
    import torch.distributed as dist
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from lowlightenvironmentvideoobjectdetection_amd import data
    from lowlightenvironmentvideoobjectdetection_amd.synthetic import seeded_weights
    nr = 10
    model = M.build_model(dict(type='FGFA', detector=retinanet_cfg(), motion=dict(type='FlowNetSimple', img_scale_factor=0.5),
                               aggregator=dict(type='EmbedAggregator', num_convs=1, channels=256, kernel_size=3)))
    model = seeded_weights(model, seed=0).to(dev).eval()
    T = nr + args.steps + args.warmup + 1
    clean = data.synth_clean_clip(min(T, 48), args.height, args.width, seed=100 + rank)
    # pad to a multiple of 128: flow_warp_feats rescales ONE flow field to every level with F.interpolate(scale_factor=...)
    # (flow.py:20-25), which only matches the P6/P7 sizes when they divide evenly (720 -> 768 rows; 736 would give 11 vs 12)
    clip, img_shape, pad_shape = data.preprocess_clip(data.darken_and_add_noise(clean, seed=200 + rank), dev, size_divisor=128)
    frames = clip[2:-2, :, :, :3].permute(0, 3, 1, 2).float().contiguous()
    meta = dict(img_shape=img_shape, pad_shape=pad_shape, ori_shape=img_shape, scale_factor=[1.0, 1.0, 1.0, 1.0], flip=False,
                img_norm_cfg=dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True))
    refs = frames[:nr].unsqueeze(0)
    # trained-detector-like score statistics, as in the default mode (SURVEY.md 8d "busy" variant): with the raw seeded head every score is
    # ~sigmoid(-2) and the per-level top-k degenerates to selecting among 10^5 ties (topk_final 436 us per frame instead of ~20)
    from lowlightenvironmentvideoobjectdetection_amd import engine, ops
    from lowlightenvironmentvideoobjectdetection_amd.synthetic import calibrate_head_on_device
    cal = engine.RetinaNetPlan(model.detector.state_dict(), NUM_CLASSES, 1, pad_shape[0], pad_shape[1], dev)
    ops.pack_input(frames[nr:nr + 1].contiguous(), 8, out=cal.x_in)
    cal.plan.run()
    torch.cuda.synchronize()
    calibrate_head_on_device(model.detector, cal, target_cls_mean=-6.0, target_cls_std=1.3)
    model.invalidate_plans()
    del cal

    def step(i):
        f = frames[(nr + i) % frames.shape[0]:(nr + i) % frames.shape[0] + 1]
        kw = dict(ref_img=[refs]) if i == 0 else {}
        return model(img=[f], img_metas=[[dict(meta, frame_id=i)]], return_loss=False, **kw)

    n = 0
    for _ in range(args.warmup):
        out = step(n); n += 1
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(n); n += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        print(json.dumps({
            'metric': 'frames/sec (FGFA: flow-guided aggregation over 10 reference frames + RetinaNet R-50+FPN) on 1280x720 video',
            'value': round(args.steps * world / dt, 3), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': 'config #4: 1 frame/step, 10-frame reference memo, EmbedAggregator on 5 FPN levels (256 ch), '
                                   f'FlowNetSimple at 0.5x, frames padded to {pad_shape[1]}x{pad_shape[0]}, eager launches', 'detections_last_step': int(sum(len(a) for a in out['bbox_results']))}}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()


def launcher_command(gpus, argv, port, script=None):
    """The command that starts `gpus` ranks of this script on one node, one rank per GPU, RCCL rendezvous on 127.0.0.1
    (the reference: mmtracking/tools/dist_train.sh / dist_test.sh -> torch.distributed.launch --nproc_per_node=$GPUS)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={int(gpus)}', '--master-addr', '127.0.0.1',
            '--master-port', str(int(port)), script or os.path.abspath(__file__)] + list(argv)


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """--gpus N > 1 outside torchrun: run the N ranks as a child process group and forward rank 0's JSON line. No GPU call has
    happened in this process yet (it never initialises the device: the children do), and nothing is exec'ed."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = launcher_command(args.gpus, argv, _free_port())
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(r.stdout)
    sys.stdout.flush()
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--frames-per-step', type=int, default=16,
                    help='consecutive frames of the clip per step (throughput mode; the 1-frame latency mode is reported beside it)')
    ap.add_argument('--clip-frames', type=int, default=32)
    ap.add_argument('--height', type=int, default=720)
    ap.add_argument('--width', type=int, default=1280)
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--pipeline', type=int, default=0,
                    help='0 (default): the two halves of a step back to back on one stream, so that the per-kernel durations of '
                         'the roofline block are those of the timed region; 1: two-stage software pipeline (FastDVDnet of step i || '
                         'detector of step i-1 on two HIP streams, +1.2 %% measured)')
    ap.add_argument('--copy-window', action='store_true',
                    help="stage every step's window in the plan's own frame buffer (round 1's two copies per step) instead of reading "
                         'it in place from the resident clip')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-latency-mode', action='store_true')
    ap.add_argument('--mode', choices=['infer', 'train', 'yolo', 'fgfa', 'insert', 'selsa', 'dff2s', 'fgfa2s'], default='infer',
                    help="'train' = BASELINE.json config #3: RetinaNet fwd + loss + bwd + gradient all-reduce + SGD, one image per GPU")
    ap.add_argument('--train-batch', type=int, default=1, help='images per GPU per training step')
    ap.add_argument('--shard', choices=['frames', 'clips'], default='frames',
                    help="infer mode, N > 1 ranks: 'frames' (default) = ONE clip of clip-frames * N frames cut into N contiguous segments, "
                         "each rank its segment + 2+2 halo frames (north_star: clips shard by frame); 'clips' = every rank its own "
                         'clip of clip-frames frames (whole-video sharding, the reference\'s DistributedVideoSampler unit)')
    ap.add_argument('--dump-launches', default=None,
                    help='write the timed plan\'s launch list (step name, device kernel, algorithmic flops / bytes) as JSON: '
                         'tools/traffic_from_pmc.py attributes the PMC bytes of every dispatch to its step with it')
    ap.add_argument('--dump-results', default=None,
                    help="write the detections of every frame of the clip, gathered on rank 0 in frame order, to this .npz ('frames' sharding)")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} '
                         f'(or without torchrun: bench.py starts the ranks itself)')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    # rehearsal of the multi-rank path on a one-GPU box: LLV_BENCH_ONE_GPU=1 puts every rank on device 0 and uses gloo
    # (RCCL refuses two ranks on one device); never set by the driver
    one_gpu = os.environ.get('LLV_BENCH_ONE_GPU', '0') == '1'
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        # set-up (synthetic clip, weight packing) runs on the host in every rank: share the cores instead of oversubscribing them
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 8
        torch.set_num_threads(max(1, min(16, ncpu // world)))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f'bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}')
        world = dist.get_world_size()

    from lowlightenvironmentvideoobjectdetection_amd import data, engine, ops
    from lowlightenvironmentvideoobjectdetection_amd.synthetic import calibrate_head_on_device

    if args.mode == 'train':
        return train_bench(args, world, rank, dev)
    if args.mode == 'yolo':
        return yolo_bench(args, world, rank, dev)
    if args.mode == 'fgfa':
        return fgfa_bench(args, world, rank, dev)
    if args.mode == 'insert':
        return insert_bench(args, world, rank, dev)
    if args.mode == 'selsa':
        return selsa_bench(args, world, rank, dev)
    if args.mode in ('dff2s', 'fgfa2s'):
        return vid2s_bench(args, world, rank, dev, args.mode)

    B, T = args.frames_per_step, args.clip_frames
    det, cleaner = build_models(seed=0)
    from lowlightenvironmentvideoobjectdetection_amd import dist_utils
    frame_sharded = args.shard == 'frames'
    if frame_sharded:
        # ONE clip of T * world frames, defined frame by frame by (seed, frame index); rank r owns the contiguous segment
        # [seg0, seg1) and reads two halo frames on either side (FastDVDnet's 5-frame window; the clip's own ends are replicated, as
        # the reference's test-time sampler clamps). Frames 0..2 travel with every rank: the Brighten gain is the clip's (taken from
        # frame 0, transforms.py:421-430) and the head calibration below runs on the window of frame 0, so every rank computes with
        # the same numbers the one-rank run uses -- the concatenated detections are bit-identical to it (tests/test_gpu_shard.py).
        T_total = T * world
        seg0, seg1 = dist_utils.shard_clip(T_total, rank, world)
        assert seg1 - seg0 == T
        lo, hi = max(0, seg0 - 2), min(T_total, seg1 + 2)

        def synth(first, n):
            return data.darken_and_add_noise(data.synth_clean_clip(n, args.height, args.width, seed=100, first=first), seed=200, first=first)
        head_n = min(3, T_total)
        u8 = torch.cat([synth(0, head_n), synth(lo, hi - lo)], 0).to(torch.uint8).to(dev)
        allf, img_shape, pad_shape = data.preprocess_clip_device(u8, replicate_ends=0)          # gain from frame 0 of the CLIP
        head, local = allf[:head_n], allf[head_n:]
        clip = local[[min(max(t, 0), T_total - 1) - lo for t in range(seg0 - 2, seg1 + 2)]].contiguous()
        calib_window = head[[0, 0, 0, min(1, head_n - 1), min(2, head_n - 1)]].contiguous()
        del allf, local, u8
    else:
        # each rank gets its own clip (clips shard across ranks; a video's frames stay on one device)
        T_total, seg0 = T, 0
        clean = data.synth_clean_clip(T, args.height, args.width, seed=100 + rank)
        dark = data.darken_and_add_noise(clean, seed=200 + rank)
        # Brighten -> Normalize -> Pad on the device kernel (llv_brighten_normalize_pad_u8), outside the timed region as in the
        # reference's benchmark (data loading excluded)
        clip, img_shape, pad_shape = data.preprocess_clip_device(dark.to(torch.uint8).to(dev))
        calib_window = None
    hp, wp = pad_shape[0], pad_shape[1]

    def make_plan(cls=engine.EnhanceDetectPlan):
        return cls(cleaner.state_dict(), det.state_dict(), NUM_CLASSES, B, hp, wp, dev,
                   detector_kwargs=dict(test_cfg=dict(det.bbox_head.test_cfg), img_shapes=[img_shape] * B),
                   use_graph=not args.no_graph)
    # one pass with the raw seeded weights, then give the head trained-detector-like statistics so that a
    # few thousand candidates pass score_thr and NMS has real work ("busy" variant)
    if calib_window is not None:
        plan = engine.EnhanceDetectPlan(cleaner.state_dict(), det.state_dict(), NUM_CLASSES, 1, hp, wp, dev,
                                        detector_kwargs=dict(test_cfg=dict(det.bbox_head.test_cfg), img_shapes=[img_shape]), use_graph=False)
        plan.den.frames.copy_(calib_window)
    else:
        plan = make_plan()
        plan.den.frames.copy_(clip[:B + 4])
    plan.plan.run()
    torch.cuda.synchronize()
    calibrate_head_on_device(det, plan.det, target_cls_mean=-6.0, target_cls_std=1.3)
    del plan
    plan = make_plan(engine.PipelinedEnhanceDetectPlan if args.pipeline else engine.EnhanceDetectPlan)
    if args.pipeline:
        plan.prepare(clip)
    else:
        plan.prepare(clip, zero_copy=not args.copy_window)      # the window is read in place: no copy inside or in front of the graph
    n_windows = T - B + 1
    if args.dump_launches and rank == 0:
        pl = plan.plan if hasattr(plan, 'plan') else None
        if pl is not None:
            with open(args.dump_launches, 'w') as f:
                json.dump([dict(name=n, kernel=pl.kernel_of.get(n, ''), flops=fl, bytes=nb) for (n, _, _, fl, nb) in pl.steps], f, indent=0)

    def run_steps(k, start=0):
        for s in range(k):
            plan.step(clip, ((start + s) * B) % n_windows)

    run_steps(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps, args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # pipelined: every timed step ran one FastDVDnet pass and one detector pass (of the window before it); the detector pass
    # of the last window follows here, outside the timed region, as the first timed step's one lay inside it
    if args.pipeline:
        plan.flush()
    n_det = sum(len(d) for d, _ in plan.results())
    # frame-sharded clip: one untimed pass over the rank's whole segment, window by window, then the only exchange of the path, the
    # gather of the per-frame detections to rank 0 in frame order (multi_gpu_test's collect step, mmtrack/apis/test.py:117-122)
    frames_gathered = None
    if frame_sharded and not args.pipeline:
        per_frame, done = [], 0
        starts = list(range(0, T - B + 1, B))
        if starts[-1] + B < T:
            starts.append(T - B)
        for t0_ in starts:
            plan.step(clip, t0_)
            res = plan.results()
            for n_ in range(B):
                if t0_ + n_ >= done:
                    per_frame.append((res[n_][0].numpy().copy(), res[n_][1].numpy().copy()))
            done = t0_ + B
        assert len(per_frame) == T
        try:
            allres = dist_utils.collect_result_dicts({'bbox_results': per_frame})
            frames_gathered = len(allres['bbox_results']) if rank == 0 else None
        except Exception as e:                          # the timed number stands on its own; report the failed gather instead of dying
            allres, frames_gathered = None, f'gather failed: {type(e).__name__}: {e}'
        if rank == 0 and args.dump_results and allres is not None:
            import numpy as np
            np.savez(args.dump_results, **{f'dets_{i}': d for i, (d, _) in enumerate(allres['bbox_results'])},
                     **{f'labels_{i}': l for i, (_, l) in enumerate(allres['bbox_results'])})
    trips = {'throughput': ops.guard_trips()}            # capacity guards of decode / NMS that fired so far (0 = consistent inputs)
    last = plan.p[plan._i & 1] if args.pipeline else plan
    # batch-1 latency mode (the reference's per-frame protocol: 3 temp1 + 1 temp2 DenBlocks per frame)
    lat = None
    stream = None
    stream_pipe = None
    stream_pipe3 = None
    if rank == 0 and world == 1 and B != 1 and not args.no_latency_mode:
        p1 = engine.EnhanceDetectPlan(cleaner.state_dict(), det.state_dict(), NUM_CLASSES, 1, hp, wp, dev,
                                      detector_kwargs=dict(test_cfg=dict(det.bbox_head.test_cfg), img_shapes=[img_shape]),
                                      use_graph=not args.no_graph)
        p1.prepare(clip, zero_copy=not args.copy_window)
        for s_ in range(args.warmup):
            p1.step(clip, s_ % T)
        torch.cuda.synchronize()
        nlat = max(args.steps, 200)          # the reference's protocol (benchmark.py:83-109): batch 1, >= 200 frames
        t1 = time.perf_counter()
        for s_ in range(nlat):
            p1.step(clip, (args.warmup + s_) % T)
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t1
        trips['latency'] = ops.guard_trips()
        lat = {'frames_per_step': 1, 'frames': nlat, 'denblocks_per_frame': 4, 'value': round(nlat / d1, 3), 'unit': 'frames/s',
               'ms_per_frame': round(d1 / nlat * 1e3, 4), 'algorithmic_gflop_per_frame': round(p1.plan.flops / 1e9, 1)}
        # streaming mode: consecutive frames of one video, the two first-stage DenBlock results a frame shares with its
        # predecessor carried over instead of recomputed (2 DenBlocks per frame; the first frame of each pass over the clip is a
        # full 3+1 step). Same kernels, bit-identical output (tests/test_gpu_models.py::test_streaming_steps_equal_full_steps).
        p1.prepare_stream()
        for s_ in range(args.warmup):
            p1.step(clip, s_ % T, carry=(s_ % T) != 0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        full_steps = 0
        for s_ in range(nlat):
            t0_ = (args.warmup + s_) % T
            full_steps += t0_ == 0
            p1.step(clip, t0_, carry=t0_ != 0)
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t1
        trips['streaming'] = ops.guard_trips()
        stream = {'frames_per_step': 1, 'frames': nlat, 'denblocks_per_frame': round((4 * full_steps + 2 * (nlat - full_steps)) / nlat, 3),
                  'value': round(nlat / d1, 3), 'unit': 'frames/s', 'ms_per_frame': round(d1 / nlat * 1e3, 4),
                  'note': 'consecutive frames of one video; first-stage DenBlock results shared with the previous frame are carried over'}
        del p1
        # the same consecutive frames as a two-stage pipeline: FastDVDnet streaming step of frame i || detector of frame i-1 on two
        # HIP streams (engine.PipelinedStreamingPlan); one frame enters per step, its detections leave one step later
        pp = engine.PipelinedStreamingPlan(cleaner.state_dict(), det.state_dict(), NUM_CLASSES, 1, hp, wp, dev,
                                           detector_kwargs=dict(test_cfg=dict(det.bbox_head.test_cfg), img_shapes=[img_shape]),
                                           use_graph=not args.no_graph)
        pp.prepare(clip)
        for s_ in range(args.warmup):
            pp.step(clip, s_ % T, carry=(s_ % T) != 0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        full_steps = 0
        for s_ in range(nlat):
            t0_ = (args.warmup + s_) % T
            full_steps += t0_ == 0
            pp.step(clip, t0_, carry=t0_ != 0)
        pp.flush()
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t1
        trips['streaming_pipelined'] = ops.guard_trips()
        stream_pipe = {'frames_per_step': 1, 'frames': nlat, 'denblocks_per_frame': round((4 * full_steps + 2 * (nlat - full_steps)) / nlat, 3),
                       'value': round(nlat / d1, 3), 'unit': 'frames/s', 'ms_per_frame': round(d1 / nlat * 1e3, 4),
                       'note': 'streaming_mode with FastDVDnet(frame i) and the detector(frame i-1) on two HIP streams; detections of a '
                               'frame are available one step later'}
        del pp
        # three stages: backbone + neck of frame i-1 and head .. NMS of frame i-2 on two more streams
        pp = engine.PipelinedStreamingPlan(cleaner.state_dict(), det.state_dict(), NUM_CLASSES, 1, hp, wp, dev,
                                           detector_kwargs=dict(test_cfg=dict(det.bbox_head.test_cfg), img_shapes=[img_shape]),
                                           use_graph=not args.no_graph, stages=3)
        pp.prepare(clip)
        for s_ in range(args.warmup):
            pp.step(clip, s_ % T, carry=(s_ % T) != 0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        full_steps = 0
        for s_ in range(nlat):
            t0_ = (args.warmup + s_) % T
            full_steps += t0_ == 0
            pp.step(clip, t0_, carry=t0_ != 0)
        pp.flush()
        pp.flush()
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t1
        trips['streaming_pipelined3'] = ops.guard_trips()
        stream_pipe3 = {'frames_per_step': 1, 'frames': nlat, 'denblocks_per_frame': round((4 * full_steps + 2 * (nlat - full_steps)) / nlat, 3),
                        'value': round(nlat / d1, 3), 'unit': 'frames/s', 'ms_per_frame': round(d1 / nlat * 1e3, 4),
                        'note': 'streaming_mode on three HIP streams: FastDVDnet(frame i), backbone + neck(frame i-1), head .. NMS(frame i-2); '
                                'detections of a frame are available two steps later'}
        del pp
    n_cand_thr = int((last.det.post.dp.scores.reshape(B, -1)[0] > 0.05).sum().item())

    out = None
    if rank == 0:
        fps = args.steps * B * world / dt
        flops_step = plan.plan.flops
        out = {
            'metric': METRIC, 'value': round(fps, 3), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {
                'workload': 'Enhance-CNN (FastDVDnet 5-frame window) + RetinaNet R-50+FPN inference + decode + NMS, '
                            f'{args.width}x{args.height} clip (padded {wp}x{hp}), {NUM_CLASSES} classes, 1xMI355X per rank',
                'clip_frames': T, 'frames_per_step': B, 'denblocks_per_frame': round((2 * B + 2) / B, 3),
                'hip_graph': not args.no_graph, 'window': 'copied' if (args.copy_window or args.pipeline) else 'read in place (offset word)',
                'mode': 'throughput' if B > 1 else 'per-frame',
                'pipeline': ('2-stage: FastDVDnet(step i) || detector(step i-1) on two HIP streams, one graph per step'
                             if args.pipeline else 'none'),
                'algorithmic_gflop_per_frame': round(flops_step / B / 1e9, 1),
                'candidates_above_score_thr': n_cand_thr, 'detections_last_step': n_det,
                'postprocess_guard_trips': ops.guard_trips(), 'postprocess_guard_trips_after_phase': trips,
                'parallelism': (f'one clip of {T_total} frames sharded by frame: {world} contiguous segment(s) of {T} frames + 2+2 halo frames, '
                                f'no data-path collective (results gathered to rank 0 after the timed region)') if frame_sharded
                               else f'clip-sharded x{world} (every rank its own clip), no data-path collective',
                'frames_gathered_on_rank0': frames_gathered,
            },
        }
        # ---- roofline of the dominant kernel: the conv kernel with the largest share of the step's time. Every launch
        # is timed with a HIP event pair on the launch stream (Plan.run_timed); flops are the algorithmic 2*MACs of
        # the unpadded convolution. The other conv kernels are listed beside it.
        if not args.no_kernel_timing:
            timed = plan.plan.run_timed(iters=5)
            kof = plan.plan.kernel_of
            if args.dump_launches:
                with open(args.dump_launches, 'w') as f:
                    json.dump([dict(name=n, kernel=kof.get(n, ''), flops=fl, bytes=nb, ms=ms) for (n, ms, fl, nb) in timed], f, indent=0)
            per = {}
            for (n, ms, fl, nb) in timed:
                if fl <= 0:
                    continue
                k = kof.get(n, 'conv').split('+')[0]
                a = per.setdefault(k, [0.0, 0.0, 0, 0.0])
                a[0] += ms; a[1] += fl; a[2] += 1; a[3] += nb
            conv_ms = sum(a[0] for a in per.values())
            conv_fl = sum(a[1] for a in per.values())
            all_ms = sum(ms for _, ms, _, _ in timed)
            dom = max(per, key=lambda k: per[k][0])
            dms, dfl, dn, dnb = per[dom]
            peak = 2500.0
            ach = dfl / (dms * 1e-3) / 1e12
            convs = [(n, ms, fl) for (n, ms, fl, _) in timed if fl > 0]
            top = max(convs, key=lambda c: c[1])
            out['roofline'] = {
                'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                'traffic': _profiled_traffic(B, dom),
                'kernel': dom, 'launches_per_step': dn, 'avg_launch_us': round(dms / dn * 1e3, 2),
                'algorithmic_gflop_per_launch': round(dfl / dn / 1e9, 2),
                'algorithmic_bytes_per_launch': round(dnb / dn),
                'share_of_step_kernel_time': round(dms / all_ms, 3),
                # the same kernel against the memory roof (its 32/64-channel layers have fewer FLOP per byte than the machine's ridge):
                # algorithmic bytes per launch / average launch duration, peak 8 TB/s
                'hbm_view': {'achieved': round(dnb / (dms * 1e-3) / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
                             'frac': round(dnb / (dms * 1e-3) / 1e9 / 8000.0, 4),
                             # the same with the bytes the PMC counters saw (profiles/traffic.json; FETCH_SIZE x2 + WRITE_SIZE per launch)
                             'counter_bytes_per_launch': _profiled_traffic(B, dom),
                             'counter_gbps': (round(_profiled_traffic(B, dom) * dn / (dms * 1e-3) / 1e9, 1)
                                              if _profiled_traffic(B, dom) else None),
                             'counter_over_algorithmic': (round(_profiled_traffic(B, dom) * dn / dnb, 3) if _profiled_traffic(B, dom) else None)},
                'all_conv_kernels': {'achieved': round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                                     'frac': round(conv_fl / (conv_ms * 1e-3) / 1e12 / peak, 4),
                                     'launches_per_step': len(convs), 'ms_per_step': round(conv_ms, 4)},
                'per_kernel': {k: {'ms_per_step': round(a[0], 4), 'tflops': round(a[1] / (a[0] * 1e-3) / 1e12, 1), 'launches': a[2]}
                               for k, a in sorted(per.items(), key=lambda kv: -kv[1][0])},
                'all_kernels_ms_per_step': round(all_ms, 4),
                'top_launch': {'name': top[0], 'ms': round(top[1], 4), 'tflops': round(top[2] / (top[1] * 1e-3) / 1e12, 2)},
            }
        if lat is not None:
            out['latency_mode'] = lat
        if stream is not None:
            out['streaming_mode'] = stream
        if stream_pipe is not None:
            out['streaming_pipelined_mode'] = stream_pipe
        if stream_pipe3 is not None:
            out['streaming_pipelined3_mode'] = stream_pipe3
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(det, cleaner, clip.cpu(), img_shape)
        print(json.dumps(out), flush=True)
    trips_end = ops.guard_trips()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if trips_end:      # a capacity guard fired: some step's detections were computed on inconsistent lists -> the run is not a measurement
        print(f'bench.py: {trips_end} post-processing capacity guard(s) fired on rank {rank}; exiting non-zero', file=sys.stderr, flush=True)
        sys.exit(3)
    return out


if __name__ == '__main__':
    main()
