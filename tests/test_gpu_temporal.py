"""GPU parity of the temporal ops (EmbedAggregator, flow warp) and the LLVOD-style ring-buffer video detector."""
import numpy as np
import pytest
import torch

from helpers import calibrate_cls, img_meta, match_detections, randomize, retinanet_cfg, smooth_image

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


@pytest.mark.parametrize('num_convs,n,c,h,w', [(1, 4, 16, 12, 20), (2, 10, 256, 23, 40), (3, 2, 32, 6, 10)])
def test_embed_aggregator_vs_oracle(dev, num_convs, n, c, h, w):
    """mmtracking/tests/test_models/test_aggregators.py:7-29 shapes + values vs the oracle."""
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from oracle.temporal import embed_aggregate
    ag = randomize(M.build_aggregator(dict(type='EmbedAggregator', num_convs=num_convs, channels=c, kernel_size=3)), seed=n)
    g = torch.Generator().manual_seed(c)
    x, ref = torch.randn(1, c, h, w, generator=g), torch.randn(n, c, h, w, generator=g)
    want = embed_aggregate(ag.state_dict(), x, ref, num_convs)
    got = ag.to(dev)(x.to(dev), ref.to(dev))
    assert got.shape == want.shape == (1, c, h, w)
    assert _rel(got.cpu(), want) < 5e-3            # fp16 embeddings; the softmax weights see cosines to ~1e-3
    with pytest.raises(AssertionError):
        ag(ref.to(dev), ref.to(dev))               # "Only support 'batch_size == 1' for x" (embed_aggregator.py:62-63)


@pytest.mark.parametrize('n,c,h,w,hf,wf', [(1, 16, 12, 20, 24, 40), (2, 64, 23, 40, 46, 80), (1, 8, 16, 16, 8, 8)])
def test_flow_warp_vs_oracle(dev, n, c, h, w, hf, wf):
    """mmtracking/tests/test_core/test_motion_utils.py:7-29 shapes + values (incl. the /W, /H grid of flow.py:31-32)."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from oracle.temporal import flow_warp_feats
    g = torch.Generator().manual_seed(h * w)
    x = torch.randn(n, c, h, w, generator=g).half().float()
    flow = torch.randn(n, 2, hf, wf, generator=g) * 3
    want = flow_warp_feats(x, flow)
    got = ops.flow_warp(x.permute(0, 2, 3, 1).contiguous().half().to(dev), flow.to(dev))
    assert _rel(got.float().cpu().permute(0, 3, 1, 2), want) < 2e-3


def test_llvod_ring_buffer_detector_vs_oracle(dev):
    """3 reference frames + key frame through the ring buffer == oracle: per-level EmbedAggregator on the oracle's
    pyramid features (key excluded from the references), RetinaHead, get_bboxes."""
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from oracle.anchors import AnchorGenerator
    from oracle.bbox import DeltaXYWHBBoxCoder, bbox2result
    from oracle.detector import DEFAULT_ANCHOR, DEFAULT_TEST_CFG, get_bboxes, retina_head_forward, retinanet_extract_feat
    from oracle.temporal import embed_aggregate
    ncls, h, w = 4, 96, 128
    dcfg = retinanet_cfg(num_classes=ncls, depth=18, in_channels=(64, 128, 256, 512), feat=32)
    model = M.build_model(dict(type='LLVOD', detector=dcfg, aggregator=dict(type='EmbedAggregator', num_convs=1, channels=32),
                               memo_len=8))
    randomize(model, seed=23, cls_bias=-1.5)
    frames = smooth_image(4, h, w, seed=24) + 0.05 * torch.randn(4, 3, h, w, generator=torch.Generator().manual_seed(1))
    calibrate_cls(model.detector, frames[3:4])
    dsd = {k: v.clone() for k, v in model.detector.state_dict().items()}
    asd = {k: v.clone() for k, v in model.aggregator.state_dict().items()}
    feats = [retinanet_extract_feat(dsd, frames[i:i + 1], 18) for i in range(4)]
    agg = [embed_aggregate(asd, feats[3][l], torch.cat([feats[i][l] for i in range(3)], 0), 1) for l in range(5)]
    cls, reg = retina_head_forward(dsd, agg, 4)
    meta = img_meta(h, w)
    meta['frame_id'] = 0
    dets, labels = get_bboxes(cls, reg, AnchorGenerator(**DEFAULT_ANCHOR), DeltaXYWHBBoxCoder(), meta['img_shape'],
                              meta['scale_factor'], ncls, DEFAULT_TEST_CFG)
    ref = bbox2result(dets, labels, ncls)
    model = model.to(dev).eval()
    out = model(img=[frames[3:4].to(dev)], img_metas=[[meta]], ref_img=[frames[:3].unsqueeze(0).to(dev)], return_loss=False)
    assert set(out) == {'bbox_results'} and len(out['bbox_results']) == ncls
    frac, tot = match_detections(out['bbox_results'], ref)
    assert tot > 10 and frac >= 0.85, f'{frac:.3f} of {tot}'
    # sliding on: a fifth frame pushes into the ring and still returns well-formed results
    meta2 = dict(meta, frame_id=1)
    out2 = model(img=[frames[0:1].to(dev)], img_metas=[[meta2]], return_loss=False)
    assert model.memo['valid'] == 5 and all(a.ndim == 2 and a.shape[1] == 5 for a in out2['bbox_results'])


@pytest.mark.parametrize('n,h,w,ih,iw', [(1, 128, 192, 128, 192), (2, 96, 160, 90, 150), (1, 184, 216, 184, 200)])
def test_flownet_simple_vs_oracle(dev, n, h, w, ih, iw):
    """FlowNetSimple (strided convs, LeakyReLU, 4x4/s2 transposed convs as conv3x3 + PixelShuffle with crop_like,
    concat through channel slices) vs the oracle; fp16 storage through ~20 layers: 2e-2 of the flow range."""
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from oracle.flownet import flownet_forward
    m = randomize(M.build_motion(dict(type='FlowNetSimple', img_scale_factor=0.5)), seed=31)
    with torch.no_grad():      # keep the refinement numerically tame (randomised ConvTranspose2d weights are not touched by randomize)
        for name, p in m.named_parameters():
            if 'deconv' in name or 'upsample_flow' in name:
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(hash(name) % 1000)) * (1.0 / (p.shape[0] * 4) ** 0.5))
    meta = img_meta(ih, iw, h, w)
    imgs = torch.cat([smooth_image(n, h, w, seed=32), smooth_image(n, h, w, seed=33)], 1)
    want = flownet_forward({k: v.clone() for k, v in m.state_dict().items()}, imgs.clone(), meta, 0.5, 5.0)
    got = m.to(dev)(imgs.to(dev), [meta])
    assert got.shape == want.shape
    assert _rel(got.cpu(), want) < 2e-2


def _video_parts(ncls=4, feat=32, seed=41):
    dcfg = retinanet_cfg(num_classes=ncls, depth=18, in_channels=(64, 128, 256, 512), feat=feat)
    return dcfg, dict(type='FlowNetSimple', img_scale_factor=0.5)


def _tame_flownet(model, scale=0.02):
    """Small flow magnitudes (a random FlowNet emits flows of hundreds of pixels, which only tests border clamping)."""
    with torch.no_grad():
        for name, p in model.motion.named_parameters():
            if 'deconv' in name or 'upsample_flow' in name:
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(len(name))) * (1.0 / (p.shape[0] * 4) ** 0.5))
            if name.startswith('predict_flow'):
                p.mul_(scale)
    model.invalidate_plans()


def _oracle_head_dets(dsd, feats, meta, ncls):
    from oracle.anchors import AnchorGenerator
    from oracle.bbox import DeltaXYWHBBoxCoder, bbox2result
    from oracle.detector import DEFAULT_ANCHOR, DEFAULT_TEST_CFG, get_bboxes, retina_head_forward
    cls, reg = retina_head_forward(dsd, feats, 4)
    d, l = get_bboxes(cls, reg, AnchorGenerator(**DEFAULT_ANCHOR), DeltaXYWHBBoxCoder(), meta['img_shape'], meta['scale_factor'],
                      ncls, DEFAULT_TEST_CFG)
    return bbox2result(d, l, ncls)


def test_dff_key_and_flow_warped_frames_vs_oracle(dev):
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from oracle.detector import retinanet_extract_feat
    from oracle.flownet import flownet_forward
    from oracle.temporal import flow_warp_feats
    ncls, h, w = 4, 128, 192
    dcfg, mcfg = _video_parts(ncls)
    model = randomize(M.build_model(dict(type='DFF', detector=dcfg, motion=mcfg, test_cfg=dict(key_frame_interval=10))), seed=43,
                      cls_bias=-1.5)
    _tame_flownet(model)
    f0 = smooth_image(1, h, w, seed=44)
    f1 = torch.roll(f0, shifts=(2, 3), dims=(2, 3)) + 0.02 * torch.randn(1, 3, h, w, generator=torch.Generator().manual_seed(2))
    calibrate_cls(model.detector, f0)
    dsd = {k: v.clone() for k, v in model.detector.state_dict().items()}
    msd = {k: v.clone() for k, v in model.motion.state_dict().items()}
    meta0, meta1 = dict(img_meta(h, w), frame_id=0), dict(img_meta(h, w), frame_id=1)
    feats0 = retinanet_extract_feat(dsd, f0, 18)
    ref0 = _oracle_head_dets(dsd, feats0, meta0, ncls)
    flow = flownet_forward(msd, torch.cat((f1, f0), 1), meta1, 0.5, 5.0)
    assert 0.05 < flow.abs().max().item() < 40
    ref1 = _oracle_head_dets(dsd, [flow_warp_feats(f, flow) for f in feats0], meta1, ncls)
    model = model.to(dev).eval()
    out0 = model(img=[f0.to(dev)], img_metas=[[meta0]], return_loss=False)
    out1 = model(img=[f1.to(dev)], img_metas=[[meta1]], return_loss=False)
    for out, ref in ((out0, ref0), (out1, ref1)):
        frac, tot = match_detections(out['bbox_results'], ref)
        assert tot > 10 and frac >= 0.8, f'{frac:.3f} of {tot}'


def test_fgfa_vs_oracle(dev):
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from oracle.detector import retinanet_extract_feat
    from oracle.flownet import flownet_forward
    from oracle.temporal import embed_aggregate, flow_warp_feats
    ncls, h, w, nr = 4, 128, 192, 3
    dcfg, mcfg = _video_parts(ncls)
    model = randomize(M.build_model(dict(type='FGFA', detector=dcfg, motion=mcfg,
                                         aggregator=dict(type='EmbedAggregator', num_convs=1, channels=32))), seed=47, cls_bias=-1.5)
    _tame_flownet(model)
    f = smooth_image(1, h, w, seed=48)
    refs = torch.cat([torch.roll(f, shifts=(i - 1, 2 * i - 2), dims=(2, 3)) + 0.02 * torch.randn(1, 3, h, w, generator=torch.Generator().manual_seed(i))
                      for i in range(nr)], 0)
    calibrate_cls(model.detector, f)
    dsd = {k: v.clone() for k, v in model.detector.state_dict().items()}
    msd = {k: v.clone() for k, v in model.motion.state_dict().items()}
    asd = {k: v.clone() for k, v in model.aggregator.state_dict().items()}
    meta = dict(img_meta(h, w), frame_id=0)
    x = retinanet_extract_feat(dsd, f, 18)
    rx = retinanet_extract_feat(dsd, refs, 18)
    flows = flownet_forward(msd, torch.cat((f.repeat(nr, 1, 1, 1), refs), 1), meta, 0.5, 5.0)
    agg = []
    for l in range(5):
        warped = flow_warp_feats(rx[l], flows)
        agg.append(embed_aggregate(asd, x[l], torch.cat((x[l], warped), 0), 1))       # fgfa.py:276-281
    ref = _oracle_head_dets(dsd, agg, meta, ncls)
    model = model.to(dev).eval()
    out = model(img=[f.to(dev)], img_metas=[[meta]], ref_img=[refs.unsqueeze(0).to(dev)], return_loss=False)
    frac, tot = match_detections(out['bbox_results'], ref)
    assert tot > 10 and frac >= 0.8, f'{frac:.3f} of {tot}'
    out2 = model(img=[refs[1:2].to(dev)], img_metas=[[dict(meta, frame_id=1)]], return_loss=False)     # memo reused
    assert all(a.ndim == 2 and a.shape[1] == 5 for a in out2['bbox_results'])


def test_fgfa_fixed_stride_vs_oracle(dev):
    """extract_feats' fixed-stride branch (fgfa.py:232-262,275-276): sliding reference window, the key frame's features
    taken from the window at num_left_ref_imgs and substituted for that warped slot; three consecutive frames."""
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from oracle.detector import retinanet_extract_feat
    from oracle.flownet import flownet_forward
    from oracle.temporal import embed_aggregate, flow_warp_feats
    ncls, h, w, nr, left, stride = 4, 128, 192, 3, 1, 2
    dcfg, mcfg = _video_parts(ncls)
    model = randomize(M.build_model(dict(type='FGFA', detector=dcfg, motion=mcfg,
                                         aggregator=dict(type='EmbedAggregator', num_convs=1, channels=32))), seed=51, cls_bias=-1.5)
    _tame_flownet(model)
    f = smooth_image(1, h, w, seed=52)
    frames = [torch.roll(f, shifts=(i, 2 * i), dims=(2, 3)) + 0.02 * torch.randn(1, 3, h, w, generator=torch.Generator().manual_seed(10 + i))
              for i in range(5)]
    calibrate_cls(model.detector, f)
    dsd = {k: v.clone() for k, v in model.detector.state_dict().items()}
    msd = {k: v.clone() for k, v in model.motion.state_dict().items()}
    asd = {k: v.clone() for k, v in model.aggregator.state_dict().items()}

    def oracle_step(img, memo_img, memo_feats, x, meta):
        flows = flownet_forward(msd, torch.cat((img.repeat(nr, 1, 1, 1), memo_img), 1), meta, 0.5, 5.0)
        agg = []
        for l in range(5):
            warped = flow_warp_feats(memo_feats[l], flows).clone()
            warped[left] = x[l][0]
            agg.append(embed_aggregate(asd, x[l], warped, 1))
        return _oracle_head_dets(dsd, agg, meta, ncls)

    model = model.to(dev).eval()
    base = dict(img_meta(h, w), frame_stride=stride, num_left_ref_imgs=left)
    # frame 0: window = frames 0..2, the key frame is window slot `left` (= frames[1])
    memo_img = torch.cat(frames[0:3], 0)
    memo_feats = list(retinanet_extract_feat(dsd, memo_img, 18))
    x = [mf[left:left + 1] for mf in memo_feats]
    meta = dict(base, frame_id=0)
    ref = oracle_step(frames[1], memo_img, memo_feats, x, meta)
    out = model(img=[frames[1].to(dev)], img_metas=[[meta]], ref_img=[memo_img.unsqueeze(0).to(dev)], return_loss=False)
    frac, tot = match_detections(out['bbox_results'], ref)
    assert tot > 10 and frac >= 0.8, f'frame 0: {frac:.3f} of {tot}'
    # frame 1: no new reference, x from the frame itself
    meta = dict(base, frame_id=1)
    x = retinanet_extract_feat(dsd, frames[2], 18)
    ref = oracle_step(frames[2], memo_img, memo_feats, x, meta)
    out = model(img=[frames[2].to(dev)], img_metas=[[meta]], return_loss=False)
    frac, tot = match_detections(out['bbox_results'], ref)
    assert tot > 10 and frac >= 0.8, f'frame 1: {frac:.3f} of {tot}'
    # frame 2: the window slides by one reference frame; x is the window's slot `left` again
    meta = dict(base, frame_id=2)
    new = frames[3]
    memo_img = torch.cat((memo_img, new), 0)[1:]
    nf = retinanet_extract_feat(dsd, new, 18)
    memo_feats = [torch.cat((mf, n_), 0)[1:] for mf, n_ in zip(memo_feats, nf)]
    x = [mf[left:left + 1] for mf in memo_feats]
    ref = oracle_step(frames[2], memo_img, memo_feats, x, meta)
    out = model(img=[frames[2].to(dev)], img_metas=[[meta]], ref_img=[new.unsqueeze(0).to(dev)], return_loss=False)
    frac, tot = match_detections(out['bbox_results'], ref)
    assert tot > 10 and frac >= 0.8, f'frame 2: {frac:.3f} of {tot}'


def test_fastdvdnet_detect_model_api_vs_oracle(dev):
    """FastDVDnetDetect through the reference's test-time call (selsa_fastdvd_darkfarm.py:262-319): every frame with its own four
    reference frames (a sliding window: consecutive frames take the streaming step), then the reference's memo behaviour (no
    ref_img after frame 0: the cached references are reused). Each result against the oracle on the same window."""
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from lowlightenvironmentvideoobjectdetection_amd.synthetic import seeded_weights
    from helpers import calibrate_cls, img_meta, match_detections, retinanet_cfg, smooth_image
    from oracle.cleaner import fastdvdnet_forward
    from oracle.detector import retinanet_simple_test
    ncls, h, w, T = 4, 64, 96, 6
    model = M.build_model(dict(type='FastDVDnetDetect', pretrains=None, detector=retinanet_cfg(num_classes=ncls),
                               cleaner=dict(type='FastDVDnet'), frozen_modules='cleaner'))
    seeded_weights(model.detector, seed=51, cls_bias=-1.5)
    seeded_weights(model.cleaner, seed=52)
    g = torch.Generator().manual_seed(53)
    clean = smooth_image(1, h, w, seed=54)
    frames = torch.cat([clean + 0.3 * torch.randn(1, 3, h, w, generator=g) for _ in range(T)], 0)
    sd_c = {k: v.detach().float().cpu() for k, v in model.cleaner.state_dict().items()}
    with torch.no_grad():
        calibrate_cls(model.detector, fastdvdnet_forward(sd_c, frames[[0, 0, 0, 1, 2]].reshape(1, 15, h, w)))
    sd_d = {k: v.detach().float().cpu() for k, v in model.detector.state_dict().items()}
    model = model.to(dev).eval()

    def window_of(t):                      # reference sampler: frames t-2 .. t+2 clamped to the video
        idx = [min(max(t + o, 0), T - 1) for o in (-2, -1, 0, 1, 2)]
        return idx

    def oracle(idx):
        with torch.no_grad():
            den = fastdvdnet_forward(sd_c, frames[idx].reshape(1, 15, h, w))
            return retinanet_simple_test(sd_d, den, img_meta(h, w), num_classes=ncls)
    tot = 0
    for t in range(T):
        idx = window_of(t)
        refs = frames[[idx[0], idx[1], idx[3], idx[4]]].unsqueeze(0).to(dev)
        meta = dict(img_meta(h, w), frame_id=t)
        res = model(return_loss=False, img=[frames[t:t + 1].to(dev)], img_metas=[[meta]], ref_img=[refs], rescale=False)['bbox_results']
        frac, n = match_detections(res, oracle(idx))
        assert n > 5 and frac >= 0.8, f'frame {t}: {frac:.2f} of {n}'
        tot += n
    plan = next(iter(model._plans.values()))
    assert plan.den.plan_stream is not None           # frames 1.. took the streaming step
    # memo behaviour: frame 0 caches its references, later frames arrive without ref_img and reuse them
    idx0 = window_of(0)
    refs0 = frames[[idx0[0], idx0[1], idx0[3], idx0[4]]].unsqueeze(0).to(dev)
    model(return_loss=False, img=[frames[0:1].to(dev)], img_metas=[[dict(img_meta(h, w), frame_id=0)]], ref_img=[refs0])
    res = model(return_loss=False, img=[frames[3:4].to(dev)], img_metas=[[dict(img_meta(h, w), frame_id=1)]])['bbox_results']
    frac, n = match_detections(res, oracle([idx0[0], idx0[1], 3, idx0[3], idx0[4]]))
    assert n > 5 and frac >= 0.8
    assert tot > 40


def test_fastdvdnet_detect_forward_train_vs_oracle(dev):
    """FastDVDnetDetect.forward_train (the dense-head form of selsa_fastdvd_darkfarm.py:64-227, row a16): noisy | clean frames stacked
    on the channel axis, cyclic 5-stacks through the frozen FastDVDnet, loss_denoising_l2 against the clean frames, detector losses
    on the denoised key frame; `.backward()` fills the detector's gradients through the device backward pass. Against the oracle:
    FastDVDnet + RetinaNet forward + retina_loss in fp32 with torch autograd."""
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from lowlightenvironmentvideoobjectdetection_amd._lib import LlvError
    from oracle import detector as OD
    from oracle.anchors import AnchorGenerator
    from oracle.cleaner import fastdvdnet_forward
    from oracle.training import retina_loss
    H, W, NC = 96, 128, 4
    dcfg = retinanet_cfg(num_classes=NC, depth=18, in_channels=(64, 128, 256, 512), stacked_convs=2, feat=64)
    dcfg['train_cfg'] = dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1),
                             allowed_border=-1, pos_weight=-1, debug=False)
    model = M.build_model(dict(type='FastDVDnetDetect', detector=dcfg, cleaner=dict(type='FastDVDnet'), frozen_modules='cleaner', loss_type='l2'))
    randomize(model.cleaner, seed=71)
    randomize(model.detector, seed=72, cls_bias=-2.0)
    g = torch.Generator().manual_seed(73)
    clean = smooth_image(5, H, W, seed=74)
    noise = clean + 0.25 * torch.randn(5, 3, H, W, generator=g)
    both = torch.cat([noise, clean], 1)                         # (5, 6, H, W): frame order [ref0, ref1, key, ref2, ref3]
    img, ref_img = both[[2]], both[[0, 1, 3, 4]].unsqueeze(0)
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=1.0)]
    gtb, gtl = [torch.tensor([[10., 12., 70., 80.], [60., 30., 120., 90.]])], [torch.tensor([1, 3])]
    # ---- oracle
    sd = {k: v.detach().clone().float() for k, v in model.state_dict().items()}
    sd_c = {k[len('cleaner.'):]: v for k, v in sd.items() if k.startswith('cleaner.')}
    sd_d = {k[len('detector.'):]: v for k, v in sd.items() if k.startswith('detector.')}
    names = [n for n, p in model.detector.named_parameters()]
    for n in names:
        sd_d[n].requires_grad_(True)
    stacks = torch.cat([torch.cat([noise[[(i + j) % 5]] for j in (-2, -1, 0, 1, 2)], 1) for i in range(5)], 0)
    with torch.no_grad():
        den = fastdvdnet_forward(sd_c, stacks)
    outs = OD.fpn_forward(sd_d, OD.resnet_forward(sd_d, den[[2]], depth=18))
    cls, reg = OD.retina_head_forward(sd_d, outs, stacked_convs=2)
    ag = AnchorGenerator(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128])
    ref = retina_loss(cls, reg, gtb, gtl, metas, ag, NC)
    ref_total = sum(ref['loss_cls']) + sum(ref['loss_bbox'])
    ref_total.backward()
    ref_den_loss = torch.nn.functional.mse_loss(den, clean).item()
    # ---- device
    model = model.to(dev)
    losses = model(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in gtb], gt_labels=[l.to(dev) for l in gtl],
                   ref_img=ref_img.to(dev), ref_img_metas=[[metas[0]] * 4], return_loss=True)
    assert set(losses) == {'loss_denoising_l2', 'loss_cls', 'loss_bbox'}
    assert abs(losses['loss_denoising_l2'].item() - ref_den_loss) < 2e-2 * ref_den_loss
    total = sum(losses['loss_cls']) + sum(losses['loss_bbox'])
    assert abs(total.item() - ref_total.item()) < 3e-2 * abs(ref_total.item())
    total.backward()
    params = dict(model.detector.named_parameters())
    checked = 0
    for k in ('bbox_head.retina_cls.weight', 'bbox_head.cls_convs.1.conv.weight', 'neck.lateral_convs.1.conv.weight', 'backbone.layer4.1.conv2.weight',
              'backbone.layer2.0.conv1.weight', 'backbone.layer3.1.bn2.weight'):
        gdev, gref = params[k].grad, sd_d[k].grad
        assert gdev is not None and gref is not None, k
        scale = gref.abs().max().item()
        assert (gdev.float().cpu() - gref).abs().max().item() < 6e-2 * scale, k
        checked += 1
    assert checked == 6 and all(p.grad is None for p in model.cleaner.parameters())
    # parse_losses / train_step go through unchanged (base.py:158-220)
    out = model.train_step(dict(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in gtb], gt_labels=[l.to(dev) for l in gtl],
                                ref_img=ref_img.to(dev), ref_img_metas=[[metas[0]] * 4]), None)
    assert set(out) == {'loss', 'log_vars', 'num_samples'} and 'loss_denoising_l2' in out['log_vars']
    for p in model.cleaner.parameters():
        p.requires_grad_(True)
    with pytest.raises(LlvError, match='freeze the cleaner'):
        model(img=img.to(dev), img_metas=metas, gt_bboxes=gtb, gt_labels=gtl, ref_img=ref_img.to(dev), return_loss=True)
