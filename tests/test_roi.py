"""RoIAlign (first piece of SURVEY.md section 8f item 4). CPU: properties pinning the oracle's restatement of mmcv.ops.roi_align;
GPU: llv_roi_align against the oracle."""
import pytest
import torch


def test_oracle_roi_align_properties():
    from oracle.roi import map_roi_levels, roi_align
    h, w = 12, 16
    const = torch.full((1, 3, h, w), 2.5)
    rois = torch.tensor([[0, 8.0, 8.0, 40.0, 36.0], [0, -20.0, -20.0, 10.0, 10.0]])
    out = roi_align(const, rois, 7, 0.25, 2, True)
    assert torch.allclose(out[0], torch.full((3, 7, 7), 2.5))                    # inside the map: the constant
    assert out[1].max() <= 2.5 + 1e-6 and out[1][:, 0, 0].abs().max() == 0      # far outside: zero samples
    # a linear ramp is reproduced exactly by bilinear sampling: every bin = the ramp at the bin centre
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    ramp = (0.5 * yy - 0.25 * xx + 1.0).view(1, 1, h, w)
    roi = torch.tensor([[0, 8.0, 12.0, 50.0, 40.0]])
    out = roi_align(ramp, roi, (4, 6), 0.25, 2, True)[0, 0]
    x1, y1, x2, y2 = [v * 0.25 - 0.5 for v in (8.0, 12.0, 50.0, 40.0)]
    for oy in range(4):
        for ox in range(6):
            cy, cx = y1 + (oy + 0.5) * (y2 - y1) / 4, x1 + (ox + 0.5) * (x2 - x1) / 6
            assert out[oy, ox].item() == pytest.approx(0.5 * cy - 0.25 * cx + 1.0, abs=1e-5)
    # not aligned: no half-pixel shift and a minimum RoI size of one pixel
    tiny = torch.tensor([[0, 20.0, 20.0, 20.0, 20.0]])
    a = roi_align(ramp, tiny, 1, 0.25, 2, False)[0, 0, 0, 0].item()
    assert a == pytest.approx(0.5 * 5.5 - 0.25 * 5.5 + 1.0, abs=1e-5)            # centre of the 1x1 box at (5, 5)
    # adaptive sampling: ceil(bin size) samples per bin
    out_a = roi_align(ramp, roi, 2, 0.25, 0, True)
    assert out_a.shape == (1, 1, 2, 2)
    # FPN level mapping (single_level_roi_extractor.py:33-53): scale 56 -> level 0, 112 -> 1, 224 -> 2, 448+ -> 3
    r = torch.tensor([[0, 0, 0, 55.0, 55.0], [0, 0, 0, 112.0, 112.0], [0, 0, 0, 224.0, 224.0], [0, 0, 0, 900.0, 900.0]])
    assert map_roi_levels(r, 4).tolist() == [0, 1, 2, 3]


@pytest.mark.gpu
@pytest.mark.parametrize('n,c,h,w,k,ps,scale,sr,aligned', [(2, 64, 23, 40, 40, 7, 1 / 16, 2, True), (1, 16, 12, 20, 25, (3, 5), 0.25, 0, True),
                                                          (3, 8, 9, 9, 30, 2, 0.5, 2, False)])
def test_device_roi_align_vs_oracle(dev, n, c, h, w, k, ps, scale, sr, aligned):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from oracle.roi import roi_align
    g = torch.Generator().manual_seed(n * 10 + c)
    feat = torch.randn(n, c, h, w, generator=g).half().float()
    ctr = torch.rand(k, 2, generator=g) * torch.tensor([w / scale, h / scale])
    wh = torch.rand(k, 2, generator=g) * torch.tensor([w / scale, h / scale]) * 0.6 + 2
    rois = torch.cat([torch.randint(0, n, (k, 1), generator=g).float(), ctr - wh / 2, ctr + wh / 2], 1)
    rois[0, 1:] = torch.tensor([-30.0, -30.0, 10.0, 10.0])                        # partly outside the map
    ref = roi_align(feat, rois, ps, scale, sr, aligned)
    out = ops.roi_align(ops.pack_input(feat.to(dev), c), rois.to(dev), ps, scale, sr, aligned, out_f32=True)
    got = out.permute(0, 3, 1, 2).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 2e-3 * max(ref.abs().max().item(), 1.0)
    out16 = ops.roi_align(ops.pack_input(feat.to(dev), c), rois.to(dev), ps, scale, sr, aligned)
    assert (out16.float().permute(0, 3, 1, 2).cpu() - ref).abs().max().item() < 5e-3 * max(ref.abs().max().item(), 1.0)


def test_single_roi_extractor_builds_from_reference_cfg():
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    # bbox_roi_extractor of mmdetection/configs/_base_/models/faster_rcnn_r50_fpn.py:52-56
    m = registry.build(dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=0), out_channels=256,
                            featmap_strides=[4, 8, 16, 32]), registry.ROI_EXTRACTORS)
    assert m.num_inputs == 4 and m.roi_layers[2].spatial_scale == 1 / 16 and m.roi_layers[0].output_size == (7, 7)
    r = torch.tensor([[0, 0, 0, 55.0, 55.0], [0, 0, 0, 900.0, 900.0]])
    assert m.map_roi_levels(r, 4).tolist() == [0, 3]
    assert torch.allclose(m.roi_rescale(r[:1], 2.0), torch.tensor([[0, -27.5, -27.5, 82.5, 82.5]]))


@pytest.mark.gpu
def test_single_roi_extractor_vs_oracle(dev):
    """Four pyramid levels, RoIs of every size: each RoI is pooled from the level its scale maps to."""
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    from oracle.roi import map_roi_levels, roi_align
    m = registry.build(dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=2), out_channels=16,
                            featmap_strides=[4, 8, 16, 32]), registry.ROI_EXTRACTORS)
    g = torch.Generator().manual_seed(3)
    feats = [torch.randn(2, 16, 64 // s * 4, 96 // s * 4, generator=g).half().float() for s in (4, 8, 16, 32)]     # 256 x 384 image
    k = 50
    ctr = torch.rand(k, 2, generator=g) * torch.tensor([384.0, 256.0])
    side = torch.exp(torch.rand(k, 1, generator=g) * 5.0 + 1.8)               # 6 .. 900 px: every pyramid level gets RoIs
    wh = side * (0.7 + 0.6 * torch.rand(k, 2, generator=g))
    rois = torch.cat([torch.randint(0, 2, (k, 1), generator=g).float(), ctr - wh / 2, ctr + wh / 2], 1)
    lv = map_roi_levels(rois, 4)
    assert len(set(lv.tolist())) == 4
    ref = torch.zeros(k, 16, 7, 7)
    for i, s in enumerate((4, 8, 16, 32)):
        sel = (lv == i).nonzero().squeeze(1)
        ref[sel] = roi_align(feats[i], rois[sel], 7, 1 / s, 2, True)
    out = m(tuple(f.to(dev) for f in feats), rois.to(dev)).cpu()
    assert out.shape == ref.shape and (out - ref).abs().max().item() < 2e-3 * ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize('blocks', [4, 0])
def test_temporal_roi_align_vs_oracle(dev, blocks):
    """TemporalRoIAlign (temporal_roi_align.py:46-207; the extractor of the authors' config) on the device -- similarity maps on the
    conv engine, llv_msra_top2_gather, 3x3 embed conv over the 7x7 RoI maps, llv_troi_attention_fuse -- against oracle/selsa.py.
    The cosines are formed from fp16-rounded unit vectors: where the oracle's 2nd and 3rd best pixels are closer than that rounding,
    the pick may differ (and with it the blended feature vector), so picks must agree for >= 97 % of the (RoI point, image) pairs and
    the values are compared on the agreeing ones; without reference maps the module is plain RoIAlign."""
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    from oracle.selsa import most_similar_roi_align, temporal_roi_align
    from oracle.roi import roi_align
    C, H, W, NREF = 64, 24, 40, 3
    ext = registry.build(dict(type='TemporalRoIAlign', num_most_similar_points=2, num_temporal_attention_blocks=blocks,
                              roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=2), out_channels=C, featmap_strides=[16]),
                         registry.ROI_EXTRACTORS)
    g = torch.Generator().manual_seed(5 + blocks)
    if blocks:
        with torch.no_grad():
            ext.embed_network.conv.weight.copy_(torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5)
            ext.embed_network.conv.bias.copy_(torch.randn(C, generator=g) * 0.1)
    feat = torch.randn(1, C, H, W, generator=g)
    ref = torch.randn(NREF, C, H, W, generator=g)
    rois = torch.tensor([[0, 10., 12., 200., 150.], [0, 300., 40., 520., 300.], [0, 0., 0., 639., 383.], [0, 100., 200., 140., 260.],
                         [0, 400., 100., 480., 140.]])
    sd = {'e.' + k: v.detach().float() for k, v in ext.state_dict().items()}
    ext = ext.to(dev)
    # (1) no reference maps: RoIAlign
    plain = ext((feat.to(dev),), rois.to(dev)).cpu()
    assert (plain - roi_align(feat, rois, 7, 1 / 16, 2, True)).abs().max().item() < 2e-3 * feat.abs().max().item()
    # (2) the picks and the blended Most Similar RoI features
    rf = roi_align(feat.half().float(), rois, 7, 1 / 16, 2, True)
    o_ref, o_idx, o_w, o_cos = most_similar_roi_align(rf, ref.half().float())
    rows = rois.shape[0] * 49
    got, idx, wts, sim = ext.most_similar_roi_align(rf.permute(0, 2, 3, 1).reshape(rows, C).contiguous().to(dev), ref.to(dev), return_picks=True)
    torch.cuda.synchronize()
    sim = sim.cpu().view(NREF, H * W, -1)[:, :, :rows].permute(0, 2, 1)                # (img, row, pixel)
    assert (sim - o_cos).abs().max().item() < 2e-3                                      # cosines of fp16-rounded unit vectors
    agree = (idx.cpu().long() == o_idx).all(-1)
    assert agree.float().mean().item() >= 0.97, agree.float().mean().item()
    assert (wts.cpu()[agree] - o_w[agree]).abs().max().item() < 2e-3
    o_rows = o_ref.permute(0, 1, 3, 4, 2).reshape(NREF, rows, C)
    assert (got.float().cpu()[agree] - o_rows[agree]).abs().max().item() < 1e-2 * o_rows.abs().max().item()
    # (3) the whole module
    out = ext((feat.to(dev),), rois.to(dev), ref_feats=(ref.to(dev),)).cpu()
    exp = temporal_roi_align(sd, 'e.', feat, rois, ref, 16, blocks)
    assert out.shape == exp.shape == (rois.shape[0], C, 7, 7)
    err = (out - exp).abs().amax(dim=1) / exp.abs().max()                               # per RoI point
    assert (err < 2e-2).float().mean().item() >= 0.9, (err < 2e-2).float().mean().item()
