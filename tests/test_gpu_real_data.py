"""Parity on REAL dark frames (second half): nine consecutive frames of the reference's
in-tree noisy low-light JPEG sequence mmtracking/demo/videos/fox_noise_n10-l025/ (tests/golden/fox_noise_frames.npz, written by
tests/golden/make_golden_fox_frames.py) through the whole device path -- Brighten / Normalize / Pad (llv_brighten_normalize_pad_u8) ->
FastDVDnet (5-frame windows) -> RetinaNet R-50+FPN -> decode -> NMS -- against the oracle on the same frames, with the bars of the
synthetic full-size test (tests/test_gpu_fullsize.py::test_full_size_frame_vs_oracle). Weights are seeded (no checkpoint exists offline),
so this checks ARITHMETIC parity on real image statistics (sensor noise, JPEG blocks, a dark exposure brightened 4x), not accuracy."""
import os

import numpy as np
import pytest
import torch

from helpers import detection_parity_report, match_detections, retinanet_cfg

pytestmark = pytest.mark.gpu
NCLS = 80
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fox_noise_frames.npz')


def test_real_dark_sequence_enhance_detect_vs_oracle(dev):
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import data, engine, ops, registry
    from lowlightenvironmentvideoobjectdetection_amd.synthetic import calibrate_head_on_device, seeded_weights
    from oracle.cleaner import fastdvdnet_forward
    from oracle.detector import retinanet_simple_test
    from oracle.preprocess import preprocess_clip
    frames = np.load(FIX)['frames_bgr']
    assert frames.shape == (9, 360, 480, 3) and frames.dtype == np.uint8
    assert 40 < frames.mean() < 90                                         # a dark exposure (the sequence's name: light 0.25)
    # ---- preprocessing: the device kernel against the numpy restatement of the reference pipeline
    clip, img_shape, pad_shape = data.preprocess_clip_device(torch.from_numpy(frames).to(dev))
    hp, wp = pad_shape[:2]
    assert (hp, wp) == (384, 480) and img_shape == (360, 480, 3)
    ref_clip = torch.from_numpy(preprocess_clip(frames))                   # (9, 384, 480, 3) fp32 RGB normalised
    got_clip = clip[2:-2, :, :, :3].float().cpu()
    assert (got_clip - ref_clip).abs().max().item() <= 2e-3 * ref_clip.abs().max().item()     # fp16 storage of exact values
    assert torch.equal(clip[0], clip[2]) and torch.equal(clip[-1], clip[-3])
    # ---- seeded models, head calibrated on the first window so that NMS has real work
    det = seeded_weights(registry.build_detector(retinanet_cfg(num_classes=NCLS)), seed=0)
    cleaner = seeded_weights(registry.build_cleaner(dict(type='FastDVDnet')), seed=1)

    def plan():
        return engine.EnhanceDetectPlan(cleaner.state_dict(), det.state_dict(), NCLS, 1, hp, wp, dev,
                                        detector_kwargs=dict(test_cfg=dict(det.bbox_head.test_cfg), img_shapes=[img_shape]), use_graph=False)
    p = plan()
    p.den.frames.copy_(clip[2:7])
    p.plan.run()
    torch.cuda.synchronize()
    calibrate_head_on_device(det, p.det, target_cls_mean=-6.0, target_cls_std=1.3)
    p = plan()
    sd_c = {k: v.detach().float().cpu() for k, v in cleaner.state_dict().items()}
    sd_d = {k: v.detach().float().cpu() for k, v in det.state_dict().items()}
    reports, den_errs = [], []
    for t in (2, 4, 6):                                                    # key frames whose 5-frame window lies inside the fixture
        p.den.frames.copy_(clip[t:t + 5])                                  # clip index = frame index + 2 (two replicated frames in front)
        p.plan.run()
        torch.cuda.synchronize()
        den = ops.unpack_output(p.den.out, 3).cpu()
        dets, labels = p.det.results()[0]
        win = ref_clip[t - 2:t + 3].permute(0, 3, 1, 2).reshape(1, 15, hp, wp).half().float().contiguous()   # the fp16 values the device read
        with torch.no_grad():
            den_ref = fastdvdnet_forward(sd_c, win)
            ref = retinanet_simple_test(sd_d, den_ref, dict(img_shape=img_shape, scale_factor=[1.0, 1.0, 1.0, 1.0]), num_classes=NCLS)
        den_errs.append((den - den_ref).abs().max().item() / den_ref.abs().max().item())
        res = [dets[labels == c].numpy() for c in range(NCLS)]
        frac, tot = match_detections(res, ref)
        rep = detection_parity_report(res, ref)
        reports.append(rep)
        print(f'fox frame {100 + t}: denoise rel err {den_errs[-1]:.2e}; {frac:.3f} of {tot} oracle detections reproduced; report {rep}')
        assert den_errs[-1] < 1e-2, den_errs
        assert tot >= 50, f'only {tot} oracle detections: the head calibration did not give NMS real work'
        assert frac >= 0.85, f'{frac:.3f} of {tot}'
        assert rep['corner_px']['p95'] < 1.0 and rep['corner_px']['max'] < 4.0 and rep['score']['p95'] < 0.01, rep
    tot = sum(r['total'] for r in reports)
    print(f'real dark frames: {sum(r["matched"] for r in reports)} of {tot} oracle detections matched over three key frames; worst corner error '
          f'{max(r["corner_px"]["max"] for r in reports):.3f} px; worst denoise error {max(den_errs):.2e}')
