"""GPU parity of the implicit-GEMM conv kernel (through the C ABI) against torch fp32 conv on the CPU.

Inputs and weights are rounded to fp16 first, so the only differences are fp32 accumulation order and the
final fp16 rounding of the output: tolerance 2e-3 relative to the output scale (fp16 eps = 9.8e-4)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _run_conv(dev, n, cin, cout, h, w, k, stride, pad, flags=0, res=None, tile=0, bias=True, seed=0):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g).half().float()
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).half().float()
    b = torch.randn(cout, generator=g) if bias else None
    pc = ops.pack_conv(wt, b, stride=stride, pad=pad).to(dev)
    cin_pad = pc.cin
    xp = torch.zeros(n, h, w, cin_pad, dtype=torch.float16)
    xp[..., :cin] = _nhwc(x).half()
    y = ops.conv2d(xp.to(dev), pc, res=res, flags=flags, tile=tile)
    ref = F.conv2d(x, wt, b, stride, pad)
    return y, ref, pc


def _check(y, ref, cout, tol=2e-3):
    got = y.float().cpu()[..., :cout].permute(0, 3, 1, 2)
    scale = ref.abs().max().item() + 1e-6
    err = (got - ref).abs().max().item() / scale
    assert err < tol, f'rel err {err}'
    if y.shape[-1] > cout:
        assert y.float().cpu()[..., cout:].abs().max().item() == 0.0


@pytest.mark.parametrize('tile', [1, 2, 3, 4, 5, 6, 7, 8, 14])
@pytest.mark.parametrize('cfg', [
    # n, cin, cout, h, w, k, stride, pad
    (1, 64, 64, 20, 33, 3, 1, 1),
    (2, 32, 96, 17, 19, 3, 2, 1),
    (1, 256, 128, 9, 14, 1, 1, 0),
    (1, 24, 90, 16, 24, 3, 1, 1),
    (1, 8, 64, 31, 37, 7, 2, 3),
    (1, 128, 36, 12, 20, 3, 1, 1),
    (1, 64, 3, 16, 16, 3, 1, 1),
])
def test_conv_matches_torch(dev, cfg, tile):
    n, cin, cout, h, w, k, s, p = cfg
    y, ref, _ = _run_conv(dev, n, cin, cout, h, w, k, s, p, tile=tile)
    _check(y, ref, cout)


def test_conv_auto_tile_and_relu(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    y, ref, _ = _run_conv(dev, 1, 64, 256, 40, 56, 3, 1, 1, flags=ops.CONV_RELU)
    _check(y, F.relu(ref), 256)


def test_conv_residual_add_relu(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(5)
    res = torch.randn(1, 24, 30, 128, generator=g).half()
    y, ref, _ = _run_conv(dev, 1, 64, 128, 24, 30, 1, 1, 0, flags=ops.CONV_RELU | ops.CONV_RES_ADD, res=res.to(dev))
    _check(y, F.relu(ref + res.float().permute(0, 3, 1, 2)), 128)


def test_conv_rsub_values(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 32, 16, 16, generator=g).half().float()
    wt = (torch.randn(3, 32, 3, 3, generator=g) / 17.0).half().float()
    pc = ops.pack_conv(wt, None, stride=1, pad=1, cout_store=8).to(dev)
    res = torch.zeros(1, 16, 16, 8, dtype=torch.float16)
    res[..., :3] = torch.randn(1, 16, 16, 3, generator=g).half()
    y = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().half().to(dev), pc, res=res.to(dev), flags=ops.CONV_RES_RSUB)
    ref = res.float().permute(0, 3, 1, 2)[:, :3] - F.conv2d(x, wt, None, 1, 1)
    _check(y, ref, 3)


def test_conv_pixel_shuffle_add(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 64, 10, 12, generator=g).half().float()
    wt = (torch.randn(128, 64, 3, 3, generator=g) / 24.0).half().float()
    pc = ops.pack_conv(wt, None, stride=1, pad=1).to(dev)
    res = torch.randn(2, 20, 24, 32, generator=g).half()
    y = ops.conv2d(_nhwc(x).half().to(dev), pc, res=res.to(dev), flags=ops.CONV_PIXSHUF | ops.CONV_RES_ADD)
    ref = F.pixel_shuffle(F.conv2d(x, wt, None, 1, 1), 2) + res.float().permute(0, 3, 1, 2)
    assert y.shape == (2, 20, 24, 32)
    _check(y, ref, 32)


def test_conv_upsample_add(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 128, 23, 42, generator=g).half().float()
    wt = (torch.randn(64, 128, 1, 1, generator=g) / 11.0).half().float()
    b = torch.randn(64, generator=g)
    pc = ops.pack_conv(wt, b).to(dev)
    coarse = torch.randn(1, 12, 21, 64, generator=g).half()
    y = ops.conv2d(_nhwc(x).half().to(dev), pc, res=coarse.to(dev), flags=ops.CONV_RES_ADD | ops.CONV_RES_UPS)
    ref = F.conv2d(x, wt, b) + F.interpolate(coarse.float().permute(0, 3, 1, 2), size=(23, 42), mode='nearest')
    _check(y, ref, 64)


def test_conv_multi_source_block_diagonal(dev):
    """FastDVDnet input conv: three NHWC8 frames (3 real channels each) read by one launch as a dense
    block-diagonal restatement of Conv2d(9, 90, 3, groups=3) (FastDVDnet.py:27-28)."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(10)
    frames = [torch.randn(1, 3, 18, 22, generator=g).half().float() for _ in range(3)]
    wg = (torch.randn(90, 3, 3, 3, generator=g) / 5.0).half().float()
    dense = torch.zeros(90, 9, 3, 3)
    for gi in range(3):
        dense[30 * gi:30 * gi + 30, 3 * gi:3 * gi + 3] = wg[30 * gi:30 * gi + 30]
    in_map = []
    for gi in range(3):
        in_map += [3 * gi, 3 * gi + 1, 3 * gi + 2, -1, -1, -1, -1, -1]
    pc = ops.pack_conv(dense, None, stride=1, pad=1, in_map=in_map, cout_store=96, n_src=3).to(dev)
    xs = []
    for f in frames:
        t = torch.zeros(1, 18, 22, 8, dtype=torch.float16)
        t[..., :3] = _nhwc(f).half()
        xs.append(t.to(dev))
    y = ops.conv2d(xs, pc, flags=ops.CONV_RELU)
    ref = F.relu(F.conv2d(torch.cat(frames, 1), wg, None, 1, 1, 1, 3))
    _check(y, ref, 90)


def test_conv_grouped_towers_and_segments(dev):
    """Two tower convs side by side (groups=2) over three pyramid levels in one launch."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(11)
    levels = [(12, 20), (6, 10), (3, 5)]
    w0 = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).half().float()
    w1 = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).half().float()
    b0, b1 = torch.randn(64, generator=g), torch.randn(64, generator=g)
    pc = ops.stack_groups([ops.pack_conv(w0, b0, 1, 1), ops.pack_conv(w1, b1, 1, 1)]).to(dev)
    xs = [torch.randn(2, 128, h, w, generator=g).half().float() for h, w in levels]
    tot = sum(2 * h * w for h, w in levels)
    xbuf = torch.cat([_nhwc(x).reshape(-1, 128) for x in xs]).half().to(dev)
    ybuf = torch.zeros(tot, 128, dtype=torch.float16, device=dev)
    segs, off = [], 0
    for h, w in levels:
        segs.append((h, w, h, w, off, off))
        off += 2 * h * w
    d = ops.make_conv_desc(pc, [xbuf], ybuf, 2, segs, 128, 128, flags=ops.CONV_RELU)
    ops.launch_conv(d)
    torch.cuda.synchronize()
    off = 0
    for (h, w), x in zip(levels, xs):
        got = ybuf[off:off + 2 * h * w].float().cpu().reshape(2, h, w, 128).permute(0, 3, 1, 2)
        ref = torch.cat([F.relu(F.conv2d(x[:, :64], w0, b0, 1, 1)), F.relu(F.conv2d(x[:, 64:], w1, b1, 1, 1))], 1)
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-3, (h, w, err)
        off += 2 * h * w


def test_conv_out_f32(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    y, ref, _ = _run_conv(dev, 1, 64, 36, 12, 20, 3, 1, 1, flags=ops.CONV_OUT_F32)
    assert y.dtype == torch.float32
    _check(y, ref, 36, tol=5e-4)


def test_conv_bad_args_raise(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from lowlightenvironmentvideoobjectdetection_amd._lib import LlvError
    pc = ops.pack_conv(torch.randn(8, 8, 3, 3), None, 1, 1).to(dev)
    x = torch.zeros(1, 8, 8, 8, dtype=torch.float16, device=dev)
    with pytest.raises(LlvError):
        ops.conv2d(x, pc, flags=ops.CONV_RES_ADD)  # residual flag without residual
    with pytest.raises(LlvError):
        ops.conv2d(x.cpu(), pc)                    # no CPU path


def test_maxpool_and_pack(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 3, 21, 30, generator=g)
    sc, sh = torch.tensor([0.5, 2.0, 1.5]), torch.tensor([0.1, -0.2, 0.3])
    p = ops.pack_input(x.to(dev), 8, sc.to(dev), sh.to(dev))
    ref = (x * sc.view(1, 3, 1, 1) + sh.view(1, 3, 1, 1)).half()
    assert torch.equal(p.cpu()[..., :3], ref.permute(0, 2, 3, 1))
    assert p.cpu()[..., 3:].abs().max() == 0
    back = ops.unpack_output(p, 3)
    assert torch.equal(back.cpu(), ref.float())
    xf = torch.randn(2, 16, 21, 30, generator=g).half()
    m = ops.maxpool3x3s2(xf.permute(0, 2, 3, 1).contiguous().to(dev))
    refm = F.max_pool2d(xf.float(), 3, 2, 1)
    assert torch.equal(m.float().cpu().permute(0, 3, 1, 2), refm)


@pytest.mark.parametrize('splits', [2, 5, 16])
def test_conv_split_k(dev, splits):
    """Small-M / long-K launch (ResNet layer4, FPN P6): K split over blockIdx.z, fp32 partial slabs, fused epilogue."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(20 + splits)
    x = torch.randn(1, 512, 6, 10, generator=g).half().float()
    wt = (torch.randn(96, 512, 3, 3, generator=g) / 68.0).half().float()
    b = torch.randn(96, generator=g)
    res = torch.randn(1, 3, 5, 96, generator=g).half()
    pc = ops.pack_conv(wt, b, stride=2, pad=1).to(dev)
    y = ops.conv2d(_nhwc(x).half().to(dev), pc, res=res.to(dev), flags=ops.CONV_RELU | ops.CONV_RES_ADD, k_splits=splits)
    ref = F.relu(F.conv2d(x, wt, b, 2, 1) + res.float().permute(0, 3, 1, 2))
    _check(y, ref, 96)


def test_conv_split_k_auto_matches_unsplit(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(31)
    x = torch.randn(1, 2048, 12, 20, generator=g).half()
    wt = (torch.randn(256, 2048, 1, 1, generator=g) / 45.0).half().float()
    pc = ops.pack_conv(wt, torch.zeros(256)).to(dev)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    d = ops.make_conv_desc(pc, [xd], torch.empty(1, 12, 20, 256, dtype=torch.float16, device=dev), 1, [(12, 20, 12, 20, 0, 0)], 2048, 256)
    assert ops.conv_workspace_bytes(d) > 0                     # the heuristic splits this one
    y_auto = ops.conv2d(xd, pc)
    y_one = ops.conv2d(xd, pc, k_splits=1)
    ref = F.conv2d(x.float(), wt)
    _check(y_auto, ref, 256)
    _check(y_one, ref, 256)


def test_conv_grouped_over_separate_sources(dev):
    """FastDVDnet input conv as the engine issues it: groups == n_src == 3, group g reads frame g (NHWC8)."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(41)
    frames = [torch.randn(2, 3, 18, 22, generator=g).half().float() for _ in range(3)]
    wg = (torch.randn(90, 3, 3, 3, generator=g) / 5.0).half().float()
    pcs = [ops.pack_conv(wg[30 * i:30 * i + 30], None, 1, 1, in_map=[0, 1, 2, -1, -1, -1, -1, -1], cout_store=32) for i in range(3)]
    pc = ops.stack_groups(pcs)
    pc.n_src, pc.src_cvec = 3, 1
    pc.to(dev)
    xs = []
    for f in frames:
        t = torch.zeros(2, 18, 22, 8, dtype=torch.float16)
        t[..., :3] = _nhwc(f).half()
        xs.append(t.to(dev))
    y = ops.conv2d(xs, pc, flags=ops.CONV_RELU)
    assert y.shape == (2, 18, 22, 96)
    ref = F.relu(F.conv2d(torch.cat(frames, 1), wg, None, 1, 1, 1, 3))
    got = y.float().cpu().reshape(2, 18, 22, 3, 32)
    assert got[..., 30:].abs().max() == 0
    got = got[..., :30].reshape(2, 18, 22, 90).permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 2e-3


@pytest.mark.parametrize('cfg', [
    # n, cin, cout, h, w, flags-kind
    (1, 32, 32, 20, 45, 'relu'),
    (2, 96, 32, 17, 33, 'relu'),
    (1, 64, 64, 9, 70, 'relu'),
    (1, 64, 128, 16, 40, 'pixshuf'),
    (1, 32, 3, 24, 31, 'rsub'),
    (1, 128, 128, 8, 32, 'res'),
    (3, 8, 20, 13, 21, 'none'),
    (1, 24, 90, 16, 24, 'relu'),
])
def test_conv3x3_halo_kernel(dev, cfg):
    """Direct (halo-in-LDS) 3x3 kernel vs torch, same tolerance as the implicit-GEMM path, and bit-compatible
    results are NOT required between the two kernels (different summation order)."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    n, cin, cout, h, w, kind = cfg
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(n, cin, h, w, generator=g).half().float()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).half().float()
    b = torch.randn(cout, generator=g)
    pc = ops.add_halo_packing(ops.pack_conv(wt, b, 1, 1, cout_store=8 if cout == 3 else None)).to(dev)
    assert pc.w_halo is not None
    cin_pad = pc.cin
    xp = torch.zeros(n, h, w, cin_pad, dtype=torch.float16)
    xp[..., :cin] = _nhwc(x).half()
    ref = F.conv2d(x, wt, b, 1, 1)
    flags, res = 0, None
    if kind == 'relu':
        flags, ref = ops.CONV_RELU, F.relu(ref)
    elif kind == 'res':
        res = torch.randn(n, h, w, pc.cout, generator=g).half()
        flags, ref = ops.CONV_RELU | ops.CONV_RES_ADD, F.relu(ref + res.float().permute(0, 3, 1, 2)[:, :cout])
    elif kind == 'rsub':
        res = torch.zeros(n, h, w, 8, dtype=torch.float16)
        res[..., :3] = torch.randn(n, h, w, 3, generator=g).half()
        flags, ref = ops.CONV_RES_RSUB, res.float().permute(0, 3, 1, 2)[:, :3] - ref
    elif kind == 'pixshuf':
        res = torch.randn(n, 2 * h, 2 * w, cout // 4, generator=g).half()
        flags, ref = ops.CONV_PIXSHUF | ops.CONV_RES_ADD, F.pixel_shuffle(ref, 2) + res.float().permute(0, 3, 1, 2)
    y = ops.conv2d(xp.to(dev), pc, res=res.to(dev) if res is not None else None, flags=flags)
    _check(y, ref, ref.shape[1])


def test_conv3x3_halo_grouped_sources(dev):
    """FastDVDnet input conv through the halo kernel: groups == n_src == 3, CK = 8."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(51)
    frames = [torch.randn(2, 3, 19, 37, generator=g).half().float() for _ in range(3)]
    wg = (torch.randn(90, 3, 3, 3, generator=g) / 5.0).half().float()
    pcs = [ops.pack_conv(wg[30 * i:30 * i + 30], None, 1, 1, in_map=[0, 1, 2, -1, -1, -1, -1, -1], cout_store=32) for i in range(3)]
    pc = ops.stack_groups(pcs)
    pc.n_src, pc.src_cvec = 3, 1
    ops.add_halo_packing(pc).to(dev)
    assert pc.w_halo is not None and pc.halo_bn == 32
    xs = []
    for f in frames:
        t = torch.zeros(2, 19, 37, 8, dtype=torch.float16)
        t[..., :3] = _nhwc(f).half()
        xs.append(t.to(dev))
    y = ops.conv2d(xs, pc, flags=ops.CONV_RELU)
    ref = F.relu(F.conv2d(torch.cat(frames, 1), wg, None, 1, 1, 1, 3))
    got = y.float().cpu().reshape(2, 19, 37, 3, 32)[..., :30].reshape(2, 19, 37, 90).permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 2e-3


@pytest.mark.parametrize('use_halo', [True, False])
def test_conv_grouped_planar_outputs_then_multi_source(dev, use_halo):
    """inc0 -> inc1 as the FastDVDnet plan issues them: three groups each write their own plane
    (y_group_stride), the next conv reads the three planes as three concatenated sources."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(61)
    n, h, w = 2, 18, 37
    frames = [torch.randn(n, 3, h, w, generator=g).half().float() for _ in range(3)]
    wg = (torch.randn(90, 3, 3, 3, generator=g) / 5.0).half().float()
    w2 = (torch.randn(32, 90, 3, 3, generator=g) / 28.0).half().float()
    pcs = [ops.pack_conv(wg[30 * i:30 * i + 30], None, 1, 1, in_map=[0, 1, 2, -1, -1, -1, -1, -1], cout_store=32) for i in range(3)]
    pc = ops.stack_groups(pcs)
    pc.n_src, pc.src_cvec = 3, 1
    in_map = []
    for gi in range(3):
        in_map += [30 * gi + j if j < 30 else -1 for j in range(32)]
    pc2 = ops.pack_conv(w2, None, 1, 1, in_map=in_map, n_src=3)
    if use_halo:
        ops.add_halo_packing(pc)
        ops.add_halo_packing(pc2)
        assert pc.w_halo is not None and pc2.w_halo is not None
    pc.to(dev)
    pc2.to(dev)
    xs = []
    for f in frames:
        t = torch.zeros(n, h, w, 8, dtype=torch.float16)
        t[..., :3] = _nhwc(f).half()
        xs.append(t.to(dev))
    planes = torch.zeros(3, n, h, w, 32, dtype=torch.float16, device=dev)
    d = ops.make_conv_desc(pc, xs, planes, n, [(h, w, h, w, 0, 0)], 8, 32, flags=ops.CONV_RELU, y_group_stride=n * h * w * 32)
    ops.launch_conv(d)
    ref1 = F.relu(F.conv2d(torch.cat(frames, 1), wg, None, 1, 1, 1, 3))
    got1 = planes.float().cpu()[..., :30].permute(1, 0, 4, 2, 3).reshape(n, 90, h, w)
    assert (got1 - ref1).abs().max().item() / ref1.abs().max().item() < 2e-3
    assert planes.float().cpu()[..., 30:].abs().max().item() == 0
    y2 = ops.conv2d([planes[0], planes[1], planes[2]], pc2, flags=ops.CONV_RELU)
    ref2 = F.relu(F.conv2d(got1, w2, None, 1, 1))
    _check(y2, ref2, 32)


@pytest.mark.parametrize('n,h,w', [(1, 16, 32), (2, 19, 45), (3, 8, 70)])
def test_fused_inputcvblock(dev, n, h, w):
    """llv_inputcvblock_fwd == Conv2d(9,90,3,groups=3)+BN+ReLU -> Conv2d(90,32,3)+BN+ReLU (FastDVDnet.py:22-36),
    with the first conv's output rounded to fp16 as the two-launch path does."""
    import ctypes as C
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from lowlightenvironmentvideoobjectdetection_amd._lib import InputCvDesc, check, lib
    g = torch.Generator().manual_seed(71 + n)
    frames = [torch.randn(n, 3, h, w, generator=g).half().float() for _ in range(3)]
    wg = (torch.randn(90, 3, 3, 3, generator=g) / 5.0).half().float()
    b0 = torch.randn(90, generator=g) * 0.1
    w2 = (torch.randn(32, 90, 3, 3, generator=g) / 28.0).half().float()
    b2 = torch.randn(32, generator=g) * 0.1
    pcs = [ops.pack_conv(wg[30 * i:30 * i + 30], b0[30 * i:30 * i + 30], 1, 1, in_map=[0, 1, 2, -1, -1, -1, -1, -1], cout_store=32)
           for i in range(3)]
    pc = ops.stack_groups(pcs)
    pc.n_src, pc.src_cvec = 3, 1
    in_map = []
    for gi in range(3):
        in_map += [30 * gi + j if j < 30 else -1 for j in range(32)]
    pc2 = ops.pack_conv(w2, b2, 1, 1, in_map=in_map, n_src=3)
    ops.add_halo_packing(pc).to(dev)
    ops.add_halo_packing(pc2).to(dev)
    xs = []
    for f in frames:
        t = torch.zeros(n, h, w, 8, dtype=torch.float16)
        t[..., :3] = _nhwc(f).half()
        xs.append(t.to(dev))
    y = torch.zeros(n, h, w, 32, dtype=torch.float16, device=dev)
    d = InputCvDesc()
    d.conv = ops.make_conv_desc(pc2, xs, y, n, [(h, w, h, w, 0, 0)], 8, 32, flags=ops.CONV_RELU)
    d.w_pre, d.bias_pre = pc.w_halo.data_ptr(), pc.bias.data_ptr()
    check(lib().llv_inputcvblock_fwd(C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'llv_inputcvblock_fwd')
    mid = F.relu(F.conv2d(torch.cat(frames, 1), wg, b0, 1, 1, 1, 3)).half().float()
    ref = F.relu(F.conv2d(mid, w2, b2, 1, 1))
    _check(y, ref, 32)


@pytest.mark.parametrize('n,h,w', [(1, 16, 32), (2, 21, 50)])
def test_fused_outputcvblock(dev, n, h, w):
    """llv_outputcvblock_fwd == in1 - Conv2d(32,3,3)(ReLU(BN(Conv2d(32,32,3)(x)))) (FastDVDnet.py:64-74,115)."""
    import ctypes as C
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from lowlightenvironmentvideoobjectdetection_amd._lib import InputCvDesc, check, lib
    g = torch.Generator().manual_seed(81 + n)
    x = torch.randn(n, 32, h, w, generator=g).half().float()
    w0 = (torch.randn(32, 32, 3, 3, generator=g) / 17.0).half().float()
    b0 = torch.randn(32, generator=g) * 0.1
    w1 = (torch.randn(3, 32, 3, 3, generator=g) / 17.0).half().float()
    in1 = torch.zeros(n, h, w, 8, dtype=torch.float16)
    in1[..., :3] = torch.randn(n, h, w, 3, generator=g).half()
    pc0 = ops.add_halo_packing(ops.pack_conv(w0, b0, 1, 1)).to(dev)
    pc1 = ops.add_halo_packing(ops.pack_conv(w1, None, 1, 1, cout_store=8)).to(dev)
    xd = _nhwc(x).half().to(dev)
    y = torch.zeros(n, h, w, 8, dtype=torch.float16, device=dev)
    d = InputCvDesc()
    d.conv = ops.make_conv_desc(pc1, [xd], y, n, [(h, w, h, w, 0, 0)], 32, 8, in1.to(dev), 8, ops.CONV_RES_RSUB)
    res_keep = in1.to(dev)
    d.conv.res = res_keep.data_ptr()
    d.w_pre, d.bias_pre = pc0.w_halo.data_ptr(), pc0.bias.data_ptr()
    check(lib().llv_outputcvblock_fwd(C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'llv_outputcvblock_fwd')
    mid = F.relu(F.conv2d(x, w0, b0, 1, 1)).half().float()
    ref = in1.float().permute(0, 3, 1, 2)[:, :3] - F.conv2d(mid, w1, None, 1, 1)
    _check(y, ref, 3)


# ---- tap-streamed 3x3 kernel (halo resident in LDS, weights streamed per tap); tiles 9 (BN=128) and 10 (BN=64)
@pytest.mark.parametrize('tile', [9, 10, 11, 12, 13])
@pytest.mark.parametrize('cfg', [
    # n, cin, cout, h, w   (3x3 / stride 1 / pad 1)
    (1, 64, 64, 20, 33),      # ragged tile edges in both directions
    (2, 128, 128, 16, 64),    # exact tiles, two images
    (1, 32, 36, 9, 40),       # one chunk; cout not a multiple of 16
    (1, 256, 200, 8, 32),     # eight chunks, two n-tiles with a ragged second one
    (3, 96, 720, 5, 7),       # tile larger than the image
    (1, 32, 32, 9, 40),       # a single chunk (the grouped loop's prologue covers two of its three tap rows)
])
def test_conv3x3_tap_kernel(dev, cfg, tile):
    n, cin, cout, h, w = cfg
    if tile == 13 and cout % 8:
        pytest.skip('the 16x32-tile kernel stores 16-byte channel runs: cout % 8 == 0 (LLV_E_UNSUPPORTED otherwise)')
    y, ref, _ = _run_conv(dev, n, cin, cout, h, w, 3, 1, 1, tile=tile)
    _check(y, ref, cout)


@pytest.mark.parametrize('tile', [9, 10, 12, 13])
def test_conv3x3_tap_segments_shared_weights(dev, tile):
    """One launch over three pyramid levels with shared weights (the RetinaNet tower layout)."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(21)
    levels = [(23, 40), (12, 20), (6, 10)]
    wt = (torch.randn(96, 64, 3, 3, generator=g) / 24.0).half().float()
    b = torch.randn(96, generator=g)
    pc = ops.pack_conv(wt, b, 1, 1).to(dev)
    xs = [torch.randn(2, 64, h, w, generator=g).half().float() for h, w in levels]
    tot = sum(2 * h * w for h, w in levels)
    xbuf = torch.cat([_nhwc(x).reshape(-1, 64) for x in xs]).half().to(dev)
    ybuf = torch.zeros(tot, 96, dtype=torch.float16, device=dev)
    segs, off = [], 0
    for h, w in levels:
        segs.append((h, w, h, w, off, off))
        off += 2 * h * w
    d = ops.make_conv_desc(pc, [xbuf], ybuf, 2, segs, 64, 96, flags=ops.CONV_RELU, tile=tile)
    ops.launch_conv(d)
    torch.cuda.synchronize()
    off = 0
    for (h, w), x in zip(levels, xs):
        got = ybuf[off:off + 2 * h * w].float().cpu().reshape(2, h, w, 96).permute(0, 3, 1, 2)
        ref = F.relu(F.conv2d(x, wt, b, 1, 1))
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-3, (h, w, err)
        off += 2 * h * w


@pytest.mark.parametrize('tile', [9, 10, 12])
def test_conv3x3_tap_pixel_shuffle_residual_and_sources(dev, tile):
    """PixelShuffle + skip add epilogue (UpBlock, FastDVDnet.py:52-61) and three concatenated 32-channel sources."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(22)
    x = torch.randn(2, 64, 10, 44, generator=g).half().float()
    wt = (torch.randn(128, 64, 3, 3, generator=g) / 24.0).half().float()
    pc = ops.pack_conv(wt, None, stride=1, pad=1).to(dev)
    res = torch.randn(2, 20, 88, 32, generator=g).half()
    y = ops.conv2d(_nhwc(x).half().to(dev), pc, res=res.to(dev), flags=ops.CONV_PIXSHUF | ops.CONV_RES_ADD, tile=tile)
    ref = F.pixel_shuffle(F.conv2d(x, wt, None, 1, 1), 2) + res.float().permute(0, 3, 1, 2)
    _check(y, ref, 32)

    srcs = [torch.randn(1, 32, 13, 37, generator=g).half().float() for _ in range(3)]
    w2 = (torch.randn(32, 96, 3, 3, generator=g) / 30.0).half().float()
    b2 = torch.randn(32, generator=g)
    pc2 = ops.pack_conv(w2, b2, 1, 1, n_src=3).to(dev)
    y2 = ops.conv2d([_nhwc(s).half().to(dev) for s in srcs], pc2, flags=ops.CONV_RELU, tile=tile)
    ref2 = F.relu(F.conv2d(torch.cat(srcs, 1), w2, b2, 1, 1))
    _check(y2, ref2, 32)


def test_conv3x3_tap_rejects_ineligible(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from lowlightenvironmentvideoobjectdetection_amd._lib import LlvError
    with pytest.raises(LlvError):
        _run_conv(dev, 1, 64, 64, 16, 16, 3, 2, 1, tile=9)      # stride 2
    with pytest.raises(LlvError):
        _run_conv(dev, 1, 24, 64, 16, 16, 3, 1, 1, tile=9)      # Cin not a multiple of 32


@pytest.mark.parametrize('tile', [0, 9, 10])
def test_conv3x3_tap_grouped_towers(dev, tile):
    """cls and reg tower convs side by side (groups=2, 2x64 input channels) over two levels: the layout of
    RetinaHead tower layers 1-3 (retina_head.py:93-114). tile=0 checks the planner picks a working path."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(23)
    levels = [(24, 40), (12, 20)]
    w0 = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).half().float()
    w1 = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).half().float()
    b0, b1 = torch.randn(64, generator=g), torch.randn(64, generator=g)
    pc = ops.stack_groups([ops.pack_conv(w0, b0, 1, 1), ops.pack_conv(w1, b1, 1, 1)]).to(dev)
    xs = [torch.randn(2, 128, h, w, generator=g).half().float() for h, w in levels]
    tot = sum(2 * h * w for h, w in levels)
    xbuf = torch.cat([_nhwc(x).reshape(-1, 128) for x in xs]).half().to(dev)
    ybuf = torch.zeros(tot, 128, dtype=torch.float16, device=dev)
    segs, off = [], 0
    for h, w in levels:
        segs.append((h, w, h, w, off, off))
        off += 2 * h * w
    d = ops.make_conv_desc(pc, [xbuf], ybuf, 2, segs, 128, 128, flags=ops.CONV_RELU, tile=tile)
    ops.launch_conv(d)
    torch.cuda.synchronize()
    off = 0
    for (h, w), x in zip(levels, xs):
        got = ybuf[off:off + 2 * h * w].float().cpu().reshape(2, h, w, 128).permute(0, 3, 1, 2)
        ref = torch.cat([F.relu(F.conv2d(x[:, :64], w0, b0, 1, 1)), F.relu(F.conv2d(x[:, 64:], w1, b1, 1, 1))], 1)
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-3, (h, w, err)
        off += 2 * h * w


def test_conv3x3_tap_pixel_shuffle_subpixel_major(dev):
    """CONV_PS_EMAJOR: the same PixelShuffle + skip-add result from weights packed sub-pixel-major (ops.pixshuf_emajor),
    OC = 32 (two sub-pixels per 64-channel workgroup) and OC = 64 (one), with an odd crop."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    g = torch.Generator().manual_seed(31)
    for cin, cout, h, w in [(64, 128, 10, 44), (128, 256, 9, 33)]:
        x = torch.randn(2, cin, h, w, generator=g).half().float()
        wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).half().float()
        bias = torch.randn(cout, generator=g)
        pc = ops.pixshuf_emajor(ops.pack_conv(wt, bias, stride=1, pad=1)).to(dev)
        res = torch.randn(2, 2 * h, 2 * w, cout // 4, generator=g).half()
        y = ops.conv2d(_nhwc(x).half().to(dev), pc, res=res.to(dev), flags=ops.CONV_PIXSHUF | ops.CONV_RES_ADD | ops.CONV_PS_EMAJOR)
        ref = F.pixel_shuffle(F.conv2d(x, wt, bias, 1, 1), 2) + res.float().permute(0, 3, 1, 2)
        assert y.shape == (2, 2 * h, 2 * w, cout // 4)
        _check(y, ref, cout // 4)
        # the 16 x 32-tile kernel's epilogue of the same name (tile 13) against the 8 x 32 kernel (tile 10): same bits, with and without
        # the skip tensor and the activation
        for flags in (ops.CONV_PIXSHUF | ops.CONV_RES_ADD | ops.CONV_PS_EMAJOR, ops.CONV_PIXSHUF | ops.CONV_PS_EMAJOR | ops.CONV_RELU,
                      ops.CONV_PIXSHUF | ops.CONV_RES_ADD | ops.CONV_PS_EMAJOR | ops.CONV_RELU):
            r = res.to(dev) if flags & ops.CONV_RES_ADD else None
            ya = ops.conv2d(_nhwc(x).half().to(dev), pc, res=r, flags=flags, tile=13)
            yb = ops.conv2d(_nhwc(x).half().to(dev), pc, res=r, flags=flags, tile=10)
            assert torch.equal(ya, yb)


@pytest.mark.parametrize('tile', [0, 1, 2, 5, 6, 7])
@pytest.mark.parametrize('cfg', [
    # n, cin, cout, h, w, k, stride, pad, dilation
    (1, 64, 64, 20, 33, 3, 1, 2, 2),        # the dilated 3x3 of a DC5 stage (pad = dilation: same size); UT mode (cin % 32 == 0)
    (2, 96, 40, 13, 17, 3, 1, 2, 2),
    (1, 24, 32, 16, 24, 3, 1, 3, 3),        # general mode (cin not a multiple of 32)
    (1, 32, 48, 19, 21, 3, 2, 2, 2),        # dilation with stride 2
    (1, 64, 16, 9, 40, 3, 1, 0, 2),         # no padding: the output shrinks by 2*dilation
])
def test_conv_dilation_matches_torch(dev, cfg, tile):
    """Dilated taps (`llv_conv_desc.dilation`, resnet.py `dilations=(1, 1, 1, 2)` of the SELSA DC5 backbones) on the implicit-GEMM
    kernels, auto selection included (the tap-streamed kernel must not take a dilated layer)."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    n, cin, cout, h, w, k, s, p, d = cfg
    g = torch.Generator().manual_seed(91)
    x = torch.randn(n, cin, h, w, generator=g).half().float()
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).half().float()
    b = torch.randn(cout, generator=g)
    pc = ops.pack_conv(wt, b, stride=s, pad=p, dilation=d).to(dev)
    xp = torch.zeros(n, h, w, pc.cin, dtype=torch.float16)
    xp[..., :cin] = _nhwc(x).half()
    y = ops.conv2d(xp.to(dev), pc, flags=ops.CONV_RELU, tile=tile)
    ref = F.relu(F.conv2d(x, wt, b, s, p, d))
    assert tuple(y.shape[1:3]) == tuple(ref.shape[2:])
    _check(y, ref, cout)


def test_conv3x3_tap_grouped_loop_is_bit_identical_to_stepwise(dev):
    """Tile 12 (one barrier per tap row, counted DMA waits) against tile 10 (one barrier per tap): same k order, same bits."""
    for (n, cin, cout, h, w) in [(2, 64, 64, 37, 70), (1, 160, 96, 16, 33), (1, 32, 64, 8, 32)]:
        ya, _, _ = _run_conv(dev, n, cin, cout, h, w, 3, 1, 1, tile=10, seed=5)
        yb, _, _ = _run_conv(dev, n, cin, cout, h, w, 3, 1, 1, tile=12, seed=5)
        assert torch.equal(ya, yb)


@pytest.mark.parametrize('cfg', [(2, 64, 64, 200, 330), (4, 64, 128, 40, 70), (3, 32, 64, 50, 90), (2, 96, 72, 33, 65)])
def test_conv3x3_tap512_is_bit_identical_to_tap_kernel(dev, cfg):
    """Tile 13 (16 x 32-pixel tile, whole chunks per load phase, two workgroups per CU) against tile 10: same k order, same bits; and
    the fp32 reference. Ragged tiles, one / two / three chunks, a ragged channel tile."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    n, cin, cout, h, w = cfg
    for relu in (0, 1):
        ya, ref, _ = _run_conv(dev, n, cin, cout, h, w, 3, 1, 1, tile=13, seed=9, flags=ops.CONV_RELU * relu)
        yb, _, _ = _run_conv(dev, n, cin, cout, h, w, 3, 1, 1, tile=10, seed=9, flags=ops.CONV_RELU * relu)
        assert torch.equal(ya, yb)
        _check(ya, ref.relu() if relu else ref, cout)


def test_conv3x3_tap512_refuses_other_epilogues(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from lowlightenvironmentvideoobjectdetection_amd._lib import LlvError
    with pytest.raises(LlvError):
        _run_conv(dev, 1, 64, 128, 16, 32, 3, 1, 1, tile=13, flags=ops.CONV_PIXSHUF)      # channel-major PixelShuffle: not on this kernel
