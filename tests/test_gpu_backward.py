"""GPU parity of the backward kernels (through the C ABI) against torch autograd on the CPU (fp32).

Inputs, weights and output gradients are rounded to fp16 first; accumulation is fp32 on both sides, so the weight
gradient differs only by summation order (tolerance 2e-3 of its scale) and the data gradient additionally by its
fp16 output rounding."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


@pytest.mark.parametrize('cfg', [
    # n, cin, cout, h, w, k, stride, pad
    (2, 64, 64, 20, 33, 3, 1, 1),
    (1, 128, 256, 12, 20, 3, 1, 1),      # 128x128 tile
    (2, 256, 64, 9, 14, 1, 1, 0),
    (1, 64, 128, 17, 19, 3, 2, 1),
    (2, 128, 256, 16, 24, 1, 2, 0),
    (1, 24, 36, 13, 11, 3, 1, 1),        # ragged channels (cin padded to 24, cout 36 in a 40-wide gradient)
])
def test_wgrad_matches_autograd(dev, cfg):
    from lowlightenvironmentvideoobjectdetection_amd import backward as B, ops
    n, cin, cout, h, w, k, s, p = cfg
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, cin, h, w, generator=g).half().float()
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).requires_grad_()
    y = F.conv2d(x, wt, None, s, p)
    dy = torch.randn(y.shape, generator=g).half().float()
    y.backward(dy)
    ho, wo = y.shape[2], y.shape[3]
    cin_pad, dy_c = ops.round_up(cin, 8), ops.round_up(cout, 8)
    xp = torch.zeros(n, h, w, cin_pad, dtype=torch.float16)
    xp[..., :cin] = _nhwc(x).half()
    dyp = torch.zeros(n, ho, wo, dy_c, dtype=torch.float16)
    dyp[..., :cout] = _nhwc(dy).half()
    w_rows, k_pad = ops.round_up(cout, 16), ops.round_up(k * k * cin_pad, 32)
    dw = torch.zeros(1, w_rows, k_pad, dtype=torch.float32, device=dev)
    B.conv_wgrad(xp.to(dev), dyp.to(dev), dw, cin_pad=cin_pad, cout=cout, w_rows=w_rows, k_pad=k_pad, kh=k, kw=k, stride=s, pad=p,
                 groups=1, batch=n, seg=(h, w, ho, wo, 0, 0), x_ld=cin_pad, dy_ld=dy_c)
    got = B.unpack_wgrad(dw[0].cpu(), cout, cin, k, k, cin_pad)
    assert _rel(got, wt.grad) < 2e-3
    # accumulation: a second call doubles it
    B.conv_wgrad(xp.to(dev), dyp.to(dev), dw, cin_pad=cin_pad, cout=cout, w_rows=w_rows, k_pad=k_pad, kh=k, kw=k, stride=s, pad=p,
                 groups=1, batch=n, seg=(h, w, ho, wo, 0, 0), x_ld=cin_pad, dy_ld=dy_c)
    assert _rel(B.unpack_wgrad(dw[0].cpu(), cout, cin, k, k, cin_pad), 2 * wt.grad) < 2e-3


@pytest.mark.parametrize('k', [3, 1])
def test_wgrad_over_pyramid_levels_in_one_call(dev, k):
    """A filter shared by the pyramid levels (RetinaHead towers, retina_head.py:93-114): the levels as segments of ONE call -- for 3x3 /
    stride 1 one launch whose pixel splits run over the 8 x 8 patches of all levels (ragged: 13 x 21, 7 x 11, 4 x 6 and 2 x 3 maps, two
    images, 72 -> 40 channels), for other filters one launch per level -- equals autograd over the levels, bias gradient included."""
    from lowlightenvironmentvideoobjectdetection_amd import backward as B
    g = torch.Generator().manual_seed(5)
    n, cin, cout = 2, 72, 40
    sizes = [(13, 21), (7, 11), (4, 6), (2, 3)]
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).requires_grad_()
    bias = torch.zeros(cout, requires_grad=True)
    xs = [torch.randn(n, cin, h, w, generator=g).half().float() for h, w in sizes]
    dys = [torch.randn(n, cout, h, w, generator=g).half().float() for h, w in sizes]
    for x, dy in zip(xs, dys):
        F.conv2d(x, wt, bias, 1, k // 2).backward(dy)
    tot = sum(n * h * w for h, w in sizes)
    xbuf = torch.zeros(tot, cin, dtype=torch.float16)
    dbuf = torch.zeros(tot, cout, dtype=torch.float16)
    segs, o = [], 0
    for (h, w), x, dy in zip(sizes, xs, dys):
        xbuf[o:o + n * h * w] = _nhwc(x).reshape(-1, cin).half()
        dbuf[o:o + n * h * w] = _nhwc(dy).reshape(-1, cout).half()
        segs.append((h, w, h, w, o, o))
        o += n * h * w
    k_pad = k * k * cin
    dw = torch.zeros(1, 48, k_pad, dtype=torch.float32, device=dev)
    db = torch.zeros(cout, dtype=torch.float32, device=dev)
    B.conv_wgrad(xbuf.to(dev), dbuf.to(dev), dw, cin_pad=cin, cout=cout, w_rows=48, k_pad=k_pad, kh=k, kw=k, stride=1, pad=k // 2, groups=1,
                 batch=n, segs=segs, x_ld=cin, dy_ld=cout, db=db)
    assert _rel(B.unpack_wgrad(dw[0].cpu(), cout, cin, k, k, cin), wt.grad) < 2e-3
    assert _rel(db.cpu(), bias.grad) < 2e-3


def test_dgrad_takes_the_relu_mask_of_the_activation(dev):
    """LLV_CONV_MASK_POS: the data gradient of a conv whose input is a ReLU output, masked by that activation in the launch's epilogue
    (accumulating into a buffer that already holds another consumer's contribution) == autograd through relu -> conv."""
    from lowlightenvironmentvideoobjectdetection_amd import backward as B, ops
    g = torch.Generator().manual_seed(9)
    n, cin, cout, h, w = 2, 64, 96, 14, 19
    pre = torch.randn(n, cin, h, w, generator=g).half().float().requires_grad_()
    x = F.relu(pre)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / 24).half().float()
    y = F.conv2d(x, wt, None, 1, 1)
    dy = torch.randn(y.shape, generator=g).half().float()
    y.backward(dy)
    wd, KH, KW, pd, fl = B.dgrad_weight(wt, 1, 1, cin)
    pcd = ops.pack_conv(wd, None, 1, pd, cout_store=wd.shape[0]).to(dev)
    other = torch.randn(n, h, w, cin, generator=g).half()          # what another consumer has already accumulated
    dx = other.clone().to(dev)
    B.conv_dgrad(_nhwc(dy).half().to(dev), pcd, dx, n, [(h, w, h, w, 0, 0)], cout, cin, 0, accumulate=True, mask=_nhwc(x.detach()).half().to(dev))
    want = other.float() * (_nhwc(x.detach()) > 0) + _nhwc(pre.grad)
    assert _rel(dx.float().cpu(), want) < 4e-3


def test_wgrad_grouped_towers(dev):
    """groups=2 over a 2x64-channel tensor (the RetinaHead twin-tower layout)."""
    from lowlightenvironmentvideoobjectdetection_amd import backward as B
    g = torch.Generator().manual_seed(4)
    n, h, w = 2, 12, 20
    x = torch.randn(n, 128, h, w, generator=g).half().float()
    w0 = (torch.randn(64, 64, 3, 3, generator=g) / 24).requires_grad_()
    w1 = (torch.randn(64, 64, 3, 3, generator=g) / 24).requires_grad_()
    y = torch.cat([F.conv2d(x[:, :64], w0, None, 1, 1), F.conv2d(x[:, 64:], w1, None, 1, 1)], 1)
    dy = torch.randn(y.shape, generator=g).half().float()
    y.backward(dy)
    dw = torch.zeros(2, 64, 576, dtype=torch.float32, device=dev)
    B.conv_wgrad(_nhwc(x).half().to(dev), _nhwc(dy).half().to(dev), dw, cin_pad=64, cout=64, w_rows=64, k_pad=576, kh=3, kw=3, stride=1,
                 pad=1, groups=2, batch=n, seg=(h, w, h, w, 0, 0), x_ld=128, dy_ld=128)
    assert _rel(B.unpack_wgrad(dw[0].cpu(), 64, 64, 3, 3, 64), w0.grad) < 2e-3
    assert _rel(B.unpack_wgrad(dw[1].cpu(), 64, 64, 3, 3, 64), w1.grad) < 2e-3


@pytest.mark.parametrize('cfg', [
    # n, cin, cout, h, w, k, stride, pad
    (2, 64, 64, 20, 33, 3, 1, 1),
    (1, 256, 128, 9, 14, 1, 1, 0),
    (1, 64, 128, 18, 22, 3, 2, 1),       # even size
    (2, 64, 96, 23, 41, 3, 2, 1),        # odd size: the shuffled gradient is cropped
    (1, 128, 256, 16, 24, 1, 2, 0),
    (1, 128, 64, 15, 21, 1, 2, 0),
])
def test_dgrad_matches_autograd(dev, cfg):
    from lowlightenvironmentvideoobjectdetection_amd import backward as B, ops
    n, cin, cout, h, w, k, s, p = cfg
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, h, w, generator=g).half().float().requires_grad_()
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).half().float()
    y = F.conv2d(x, wt, None, s, p)
    dy = torch.randn(y.shape, generator=g).half().float()
    y.backward(dy)
    ho, wo = y.shape[2], y.shape[3]
    wd, KH, KW, pd, flags = B.dgrad_weight(wt, s, p, cin)
    pcd = ops.pack_conv(wd, None, 1, pd, cout_store=wd.shape[0]).to(dev)
    dx = torch.zeros(n, h, w, cin, dtype=torch.float16, device=dev)
    B.conv_dgrad(_nhwc(dy).half().to(dev), pcd, dx, n, [(h, w, ho, wo, 0, 0)], cout, cin, flags, accumulate=False)
    got = dx.float().cpu().permute(0, 3, 1, 2)
    assert _rel(got, x.grad) < 3e-3
    # accumulate on top of an existing gradient
    B.conv_dgrad(_nhwc(dy).half().to(dev), pcd, dx, n, [(h, w, ho, wo, 0, 0)], cout, cin, flags, accumulate=True)
    assert _rel(dx.float().cpu().permute(0, 3, 1, 2), 2 * x.grad) < 4e-3


def test_relu_bias_upsample_sgd(dev):
    from lowlightenvironmentvideoobjectdetection_amd import backward as B
    g = torch.Generator().manual_seed(6)
    dy = torch.randn(2, 9, 11, 64, generator=g).half()
    dy2 = torch.randn(2, 9, 11, 64, generator=g).half()
    y = torch.randn(2, 9, 11, 64, generator=g).half()
    dx = torch.empty_like(dy, device=dev)
    B.relu_bwd(dy.to(dev), y.to(dev), dx, dy2.to(dev))
    ref = ((dy.float() + dy2.float()).half().float() * (y.float() > 0)).half()
    assert torch.equal(dx.cpu(), ref)
    out = torch.zeros(64, dtype=torch.float32, device=dev)
    B.bias_grad(dy.to(dev), 2 * 9 * 11, 64, 64, out)
    assert _rel(out.cpu(), dy.float().sum((0, 1, 2))) < 1e-5
    # nearest-upsample backward at a non-integer ratio (23x40 <- 12x20, fpn.py:186-195)
    coarse = torch.randn(1, 8, 12, 20, generator=g).requires_grad_()
    fine = F.interpolate(coarse, size=(23, 40), mode='nearest')
    gf = torch.randn(fine.shape, generator=g).half().float()
    fine.backward(gf)
    dc = torch.zeros(1, 12, 20, 8, dtype=torch.float16, device=dev)
    B.upsample_add_bwd(gf.permute(0, 2, 3, 1).contiguous().half().to(dev), dc, accumulate=False)
    assert _rel(dc.float().cpu().permute(0, 3, 1, 2), coarse.grad) < 2e-3
    # SGD with momentum and weight decay against torch.optim.SGD, two steps
    w0 = torch.randn(1000, generator=g)
    p = torch.nn.Parameter(w0.clone())
    opt = torch.optim.SGD([p], lr=0.01, momentum=0.9, weight_decay=1e-4)
    wd, md = w0.clone().to(dev), torch.zeros(1000, device=dev)
    for step in range(2):
        gr = torch.randn(1000, generator=g)
        p.grad = gr.clone()
        opt.step()
        B.sgd_momentum(wd, (gr * 512).to(dev), md, 0.01, 0.9, 1e-4, 1.0 / 512, step == 0)
    assert _rel(wd.cpu(), p.detach()) < 1e-6


@pytest.mark.parametrize('cfg', [(2, 64, 64, 14, 19, 2), (1, 128, 96, 9, 12, 2), (1, 64, 64, 11, 13, 3)])
def test_dilated_conv_backward_matches_autograd(dev, cfg):
    """3x3 / stride 1 / padding = dilation (the conv2 of a DC5 stage, resnet.py:151-167 with dilations (1, 1, 1, 2)): weight gradient
    (llv_conv_desc.dilation read by the wgrad kernel's tap addressing) and data gradient (the flipped filter on the forward kernel with
    the same dilation and padding dil*(k-1) - pad)."""
    from lowlightenvironmentvideoobjectdetection_amd import backward as B, ops
    n, cin, cout, h, w, dil = cfg
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, cin, h, w, generator=g).half().float().requires_grad_()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).half().float().requires_grad_()
    y = F.conv2d(x, wt, None, 1, dil, dil)
    assert y.shape[2:] == x.shape[2:]
    dy = torch.randn(y.shape, generator=g).half().float()
    y.backward(dy)
    w_rows, k_pad = ops.round_up(cout, 16), ops.round_up(9 * cin, 32)
    dw = torch.zeros(1, w_rows, k_pad, dtype=torch.float32, device=dev)
    B.conv_wgrad(_nhwc(x.detach()).half().to(dev), _nhwc(dy).half().to(dev), dw, cin_pad=cin, cout=cout, w_rows=w_rows, k_pad=k_pad, kh=3, kw=3,
                 stride=1, pad=dil, groups=1, batch=n, seg=(h, w, h, w, 0, 0), x_ld=cin, dy_ld=cout, dilation=dil)
    assert _rel(B.unpack_wgrad(dw[0].cpu(), cout, cin, 3, 3, cin), wt.grad) < 2e-3
    wd, KH, KW, pd, flags = B.dgrad_weight(wt.detach(), 1, 1, cin)                    # flipped, in/out-transposed filter (geometry for pad 1)
    pcd = ops.pack_conv(wd, None, 1, dil * 2 - dil, cout_store=wd.shape[0], dilation=dil).to(dev)
    dx = torch.zeros(n, h, w, cin, dtype=torch.float16, device=dev)
    B.conv_dgrad(_nhwc(dy).half().to(dev), pcd, dx, n, [(h, w, h, w, 0, 0)], cout, cin, flags, accumulate=False)
    assert _rel(dx.float().cpu().permute(0, 3, 1, 2), x.grad) < 3e-3


@pytest.mark.parametrize('n,h,w,c', [(1, 9, 11, 8), (2, 64, 96, 64), (1, 33, 20, 16)])
def test_maxpool_bwd_matches_autograd(dev, n, h, w, c):
    """llv_maxpool3x3s2_nhwc_bwd vs torch autograd over F.max_pool2d(3, 2, 1) on a post-ReLU map (many tied zeros: the first maximum in
    scan order takes the gradient, ATen's rule), gradients ADDED into dx."""
    import ctypes as C
    import torch.nn.functional as F
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from lowlightenvironmentvideoobjectdetection_amd._lib import check, lib
    g = torch.Generator().manual_seed(n * 1000 + h)
    x = torch.relu(torch.randn(n, c, h, w, generator=g)).half()
    x = (x * 4).round() / 4                                    # coarse values: ties between non-zero neighbours too
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    dy = torch.randn(n, c, ho, wo, generator=g).half()
    xr = x.float().clone().requires_grad_(True)
    (F.max_pool2d(xr, 3, 2, 1) * dy.float()).sum().backward()
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    dx = torch.zeros(n, h, w, c, dtype=torch.float16, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib().llv_maxpool3x3s2_nhwc_bwd(ops._ptr(xd), ops._ptr(dyd), ops._ptr(dx), n, h, w, c, c, c, c, st), 'llv_maxpool3x3s2_nhwc_bwd')
    got = dx.float().permute(0, 3, 1, 2).cpu()
    assert (got - xr.grad).abs().max().item() < 4e-3
    assert ((got != 0) == (xr.grad.half() != 0)).all()
    check(lib().llv_maxpool3x3s2_nhwc_bwd(ops._ptr(xd), ops._ptr(dyd), ops._ptr(dx), n, h, w, c, c, c, c, st), 'llv_maxpool3x3s2_nhwc_bwd')
    assert (dx.float().permute(0, 3, 1, 2).cpu() - 2 * xr.grad).abs().max().item() < 8e-3


def test_relu_res_bwd_equals_the_two_passes(dev):
    """llv_relu_res_bwd (the block output relu(v + identity): mask and the identity's gradient in one pass) == llv_relu_bwd twice, bit for bit,
    and == autograd of relu(v + r) in fp32 up to the fp16 rounding of the stored sum."""
    from lowlightenvironmentvideoobjectdetection_amd import backward as B
    g = torch.Generator().manual_seed(5)
    n = 2 * 23 * 40 * 64
    y = torch.relu(torch.randn(n, generator=g)).half().to(dev)
    dy = torch.randn(n, generator=g).half().to(dev)
    gres0 = torch.randn(n, generator=g).half().to(dev)
    dy_a, gres_a = dy.clone(), gres0.clone()
    B.relu_bwd(dy_a, y, dy_a)
    B.relu_bwd(gres_a, None, gres_a, dy2=dy_a)
    dy_b, gres_b = dy.clone(), gres0.clone()
    B.relu_res_bwd(dy_b, y, gres_b)
    assert torch.equal(dy_a, dy_b) and torch.equal(gres_a, gres_b)
    want = dy.float() * (y.float() > 0)
    assert torch.equal(dy_b.float(), want)
    assert (gres_b.float() - (gres0.float() + want)).abs().max().item() < 4e-3
