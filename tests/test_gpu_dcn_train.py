"""Training of the authors' DCN-based temporal fusion: the gradients of mmcv's
modulated_deform_conv2d (input, offsets, mask, weight, bias: llv_modulated_deform_col2im + the 1x1 data / weight gradient kernels over
the column tensor), llv_convlstm_gate_bwd, and DenoisingAggregator / Denoising2Aggergator.forward_train (dev_autograd.py) -- every
parameter gradient and the input gradient against autograd over the fp32 oracle (oracle/dcn.py), with the three per-tensor metrics of
helpers.grad_metrics; the offset branch also against central differences of the DEVICE forward."""
import pytest
import torch

from helpers import grad_metrics, randomize

pytestmark = pytest.mark.gpu


def _report(tag, rows, e_big, cos_big, l2_big, top):
    """rows of (name, max-abs error / own scale, cosine, relative L2, own scale). A tensor whose oracle gradient is below 1e-5 of the
    largest is analytically zero (a bias in front of the softmax over frames: the same number is added to every frame's logit and cancels;
    the oracle's own value there is rounding noise): the device's must be equally negligible, and it is left out of the relative metrics."""
    zero = [r for r in rows if r[4] < 1e-5 * top]
    rows[:] = [r for r in rows if r[4] >= 1e-5 * top]
    for k, e, c, l2, s in zero:
        assert e * s < 2e-3 * top, (k, e * s, top)
    if zero:
        print(tag, 'analytically zero gradients (cancel in the softmax over frames):', [k for k, *_ in zero])
    rows.sort(key=lambda t: -t[1])
    print(tag, 'largest max-abs errors:', [(k, round(e, 3), round(c, 5), round(l2, 3)) for k, e, c, l2, s in rows[:6]])
    print(tag, 'lowest cosines:', [(k, round(c, 5)) for k, e, c, l2, s in sorted(rows, key=lambda t: t[2])[:4]])
    for k, e, c, l2, s in rows:
        big = s > 1e-3 * top
        assert e < (e_big if big else 3 * e_big), (k, e, s)
        assert c > (cos_big if big else 1 - 10 * (1 - cos_big)), (k, c, s)
        assert l2 < (l2_big if big else 3 * l2_big), (k, l2, s)


@pytest.mark.parametrize('ci,co,k,stride,relu,bn', [(64, 64, 3, 1, True, False), (64, 128, 3, 2, False, False), (128, 64, 1, 1, True, True),
                                                   (96, 24, 3, 1, False, True), (8, 64, 7, 2, True, True)])
def test_dev_conv2d_gradients_vs_autograd(dev, ci, co, k, stride, relu, bn):
    """The building block of dev_autograd.py, one convolution (+ folded eval-mode BatchNorm, + ReLU): output and d(<y, R>)/d(x, weight, bias,
    gamma, beta) against torch autograd in fp32 on the fp16-rounded input. One op deep, so the bounds are those of fp16 rounding alone."""
    import torch.nn as nn
    import torch.nn.functional as F
    from lowlightenvironmentvideoobjectdetection_amd import dev_autograd as D
    g = torch.Generator().manual_seed(ci * 7 + co + k)
    n, h, w = 2, 14, 18
    c_real = 3 if ci == 8 else ci                          # the stem: 3 real channels in an 8-channel input
    conv = nn.Conv2d(c_real, co, k, stride=stride, padding=k // 2, bias=not bn)
    norm = nn.BatchNorm2d(co).eval() if bn else None
    with torch.no_grad():
        # fp16-exact weights (as the input): the device then computes the same pre-activations as fp32 autograd up to summation order, and
        # the ReLU masks agree -- with fp32 weights the rounding to fp16 flips the sign of ~5e-4 of the pre-activations (those within 1e-3 of
        # zero), each a full-size gradient error: a 2 % relative L2 difference per ReLU layer that is a property of fp16 storage, not of
        # the kernels (measured: first run of this test)
        conv.weight.copy_((torch.randn(conv.weight.shape, generator=g) * (2.0 / (c_real * k * k)) ** 0.5).half().float())
        if conv.bias is not None:
            conv.bias.copy_(torch.randn(co, generator=g) * 0.2)
        if bn:
            norm.weight.copy_(torch.rand(co, generator=g) + 0.5); norm.bias.copy_(torch.randn(co, generator=g) * 0.2)
            norm.running_mean.copy_(torch.randn(co, generator=g) * 0.2); norm.running_var.copy_(torch.rand(co, generator=g) + 0.5)
    x = torch.randn(n, c_real, h, w, generator=g).half().float()
    xr = x.clone().requires_grad_()
    y = conv(xr)
    if bn:
        y = norm(y)
    if relu:
        y = F.relu(y)
    R = torch.randn(y.shape, generator=g)
    (y * R).sum().backward()
    ref = dict(x=xr.grad, weight=conv.weight.grad, bias=None if conv.bias is None else conv.bias.grad,
               gamma=norm.weight.grad if bn else None, beta=norm.bias.grad if bn else None)
    for m_ in (conv, norm):
        if m_ is not None:
            m_.zero_grad()
            m_.to(dev)
    xd = x.to(dev).requires_grad_()
    yd = D.from_dev(D.conv2d(D.to_dev(xd), conv, relu=relu, bn=norm), co)
    err = (yd.detach().cpu() - y.detach()).abs().max().item() / y.detach().abs().max().item()
    assert err < 2e-3, err
    (yd * R.to(dev)).sum().backward()
    got = dict(x=xd.grad, weight=conv.weight.grad, bias=None if conv.bias is None else conv.bias.grad,
               gamma=norm.weight.grad if bn else None, beta=norm.bias.grad if bn else None)
    rows = [(k_,) + grad_metrics(got[k_], ref[k_]) for k_ in ref if ref[k_] is not None]
    print(f'conv {ci}->{co} k{k} s{stride}: forward {err:.1e};', [(k_, round(e, 4), round(1 - c, 7), round(l2, 4)) for k_, e, c, l2, s_ in rows])
    # a folded BatchNorm scale makes the packed weights inexact in fp16 again: with a ReLU behind it a few masks flip (see above)
    flips = relu and bn
    for k_, e, c, l2, s_ in rows:
        assert e < (8e-2 if flips else 1e-2) and c > (0.9995 if flips else 0.99999) and l2 < (3e-2 if flips else 3e-3), (k_, e, c, l2)


@pytest.mark.parametrize('n,c,co,h,w,dg', [(2, 64, 64, 13, 17, 8), (1, 32, 24, 9, 20, 4)])
def test_modulated_deform_conv_gradients_vs_oracle(dev, n, c, co, h, w, dg):
    """d(<y, R>)/d(x, offsets, mask logits, weight, bias) of the deformable convolution. Offsets of sigma 1.5 px (some samples leave the
    image: the border rule of the gradient), mask logits O(1)."""
    from lowlightenvironmentvideoobjectdetection_amd import dev_autograd as D
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from oracle.dcn import modulated_deform_conv2d
    g = torch.Generator().manual_seed(n * 100 + c)
    K = 9
    x = torch.randn(n, c, h, w, generator=g).half().float().requires_grad_()
    wgt = (torch.randn(co, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5).half().float().requires_grad_()
    bias = (torch.randn(co, generator=g) * 0.1).requires_grad_()
    om = torch.cat([torch.randn(n, 2 * dg * K, h, w, generator=g) * 1.5, torch.randn(n, dg * K, h, w, generator=g)], 1).requires_grad_()
    R = torch.randn(n, co, h, w, generator=g)
    ref = modulated_deform_conv2d(x, om[:, :2 * dg * K], torch.sigmoid(om[:, 2 * dg * K:]), wgt, bias, 1, 1, 1, 1, dg)
    (ref * R).sum().backward()
    xd = x.detach().to(dev).requires_grad_()
    omd = om.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_()
    wd, bd = wgt.detach().to(dev).requires_grad_(), bias.detach().to(dev).requires_grad_()
    x16 = D.to_dev(xd)
    y16 = D._Mdcn.apply(x16, omd, wd, bd, dg, 1)
    y = D.from_dev(y16, co)
    assert (y.detach().cpu() - ref.detach()).abs().max().item() < 3e-3 * ref.abs().max().item()
    (y * R.to(dev)).sum().backward()
    rows = [('x',) + grad_metrics(xd.grad, x.grad),
            ('om',) + grad_metrics(omd.grad.permute(0, 3, 1, 2) / D.GRAD_SCALE, om.grad),       # om enters the op directly: its gradient is still scaled
            ('offsets',) + grad_metrics(omd.grad.permute(0, 3, 1, 2)[:, :2 * dg * K] / D.GRAD_SCALE, om.grad[:, :2 * dg * K]),
            ('mask',) + grad_metrics(omd.grad.permute(0, 3, 1, 2)[:, 2 * dg * K:] / D.GRAD_SCALE, om.grad[:, 2 * dg * K:]),
            ('weight',) + grad_metrics(wd.grad, wgt.grad), ('bias',) + grad_metrics(bd.grad, bias.grad)]
    _report('mdcn', rows, 2e-2, 0.9995, 1.5e-2, max(r[4] for r in rows))
    # the offset branch against central differences of the DEVICE forward: move every offset by +-eps along its own gradient direction
    go = omd.grad.detach()[..., :2 * dg * K]
    direction = torch.zeros_like(omd)
    direction[..., :2 * dg * K] = go / go.norm()
    analytic = float((omd.grad.detach() / D.GRAD_SCALE * direction).sum())
    eps = 0.05
    vals = []
    with torch.no_grad():
        pc = ops.pack_dcn_weight(wgt.detach(), bias.detach()).to(dev)
        for sgn in (1.0, -1.0):
            out = ops.modulated_deform_conv2d(x16.detach(), (omd.detach() + sgn * eps * direction).contiguous(), pc, 3, 3, 1, 1, 1, dg)
            vals.append(float((ops.unpack_output(out, co) * R.to(dev)).sum()))
    fd = (vals[0] - vals[1]) / (2 * eps)
    print(f'offset branch: analytic directional derivative {analytic:.4f}, central differences on the device {fd:.4f}')
    assert abs(fd - analytic) < 4e-2 * abs(analytic), (fd, analytic)


def test_convlstm_gate_backward_vs_autograd(dev):
    """llv_convlstm_gate_bwd over two chained frames (dark_resnet.py:226-246,267-269): d(loss)/d(gates of both frames, c_0, scale, shift)
    against autograd over the same formulas in fp32."""
    from lowlightenvironmentvideoobjectdetection_amd import dev_autograd as D
    g = torch.Generator().manual_seed(3)
    hh, ww, p = 6, 10, 32
    g1, g2 = [(torch.randn(1, hh, ww, 4 * p, generator=g) * 1.5).half().float().requires_grad_() for _ in range(2)]
    c0 = torch.randn(hh * ww, p, generator=g).requires_grad_()
    scale, shift = (torch.rand(p, generator=g) + 0.5).requires_grad_(), (torch.randn(p, generator=g) * 0.2).requires_grad_()
    Ra, Rh = torch.randn(2, 1, hh, ww, p, generator=g), torch.randn(1, hh, ww, p, generator=g)

    def ref_step(gt, c):
        i, f, o, gg = [torch.sigmoid(t) for t in gt.view(hh * ww, 4, p).unbind(1)]
        cn = f * c + i * gg
        hv = o * torch.tanh(cn)
        return hv, torch.relu(hv * scale + shift), cn
    h1, a1, c1 = ref_step(g1, c0)
    h2, a2, c2 = ref_step(g2, c1)
    ((a1.view(1, hh, ww, p) * Ra[0]).sum() + (a2.view(1, hh, ww, p) * Ra[1]).sum() + (h2.view(1, hh, ww, p) * Rh).sum() + (h1 * 0.5).sum()).backward()
    dv = lambda t: t.detach().to(dev).requires_grad_()
    g1d, g2d, c0d, sd_, shd = dv(g1), dv(g2), dv(c0), dv(scale), dv(shift)
    S = D.GRAD_SCALE
    hh1, aa1, cc1 = D.lstm_gate(g1d.half(), c0d, sd_, shd)
    hh2, aa2, cc2 = D.lstm_gate(g2d.half(), cc1, sd_, shd)
    assert (aa2.float().cpu().view(-1, p) - a2.detach()).abs().max().item() < 2e-3
    loss = (aa1.float() * Ra[0].to(dev)).sum() + (aa2.float() * Ra[1].to(dev)).sum() + (hh2.float() * Rh.to(dev)).sum() + (hh1.float() * 0.5).sum()
    (loss * S).backward()                       # the module boundary's factor, applied by hand here
    rows = [('gates1',) + grad_metrics(g1d.grad / S, g1.grad), ('gates2',) + grad_metrics(g2d.grad / S, g2.grad),
            ('c0',) + grad_metrics(c0d.grad / S, c0.grad), ('scale',) + grad_metrics(sd_.grad, scale.grad), ('shift',) + grad_metrics(shd.grad, shift.grad)]
    _report('lstm', rows, 1e-2, 0.9999, 5e-3, max(r[4] for r in rows))


class _Round16(torch.autograd.Function):
    """What fp16 storage does to the oracle: the activation rounded to fp16 on the way forward, its gradient (times the device's loss
    scale) rounded to fp16 on the way back."""

    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return (g * 512.0).half().float() / 512.0


def _rounding_floor(run_oracle, names):
    """The oracle's own gradients with fp16 storage emulated -- every convolution's weights rounded to fp16 (what the packed weights hold),
    its output and the gradient of its output rounded to fp16 (what the activation buffers hold) -- against its fp32 gradients: the part
    of a device-vs-oracle difference that is fp16 storage, not arithmetic. Most of it is ReLU masks: a pre-activation within ~1e-3 of
    zero changes sign under the weight rounding, and each such element is a full-size gradient error."""
    import oracle.dcn as OD
    exact = run_oracle(False)
    orig = OD._conv
    OD._conv = lambda *a, **k: _Round16.apply(orig(*a, **k))
    try:
        rounded = run_oracle(True)
    finally:
        OD._conv = orig
    return {k: grad_metrics(rounded[k], exact[k]) for k in names if exact[k] is not None}


def _conditioned_aggregator(**kw):
    from test_dcn import _aggregator
    m = _aggregator(None, **kw)
    with torch.no_grad():
        if hasattr(m, 'taf'):
            for e in m.taf.emb_conv:
                e.weight.mul_(0.3)            # a softmax over the frames that does not saturate: every frame keeps a gradient
    return m


@pytest.mark.parametrize('kw,n,h,w', [
    (dict(in_channels=64, mid_channels=64, num_blocks=1, num_layers=2, channel_growth=32, emb_nums=2), 3, 12, 16),
    (dict(in_channels=32, mid_channels=64, with_rdb=False, with_taf=True, emb_nums=1), 2, 10, 12),
    (dict(in_channels=64, mid_channels=64, num_layers=2, with_rdb=True, with_taf=False), 2, 10, 10),
])
def test_denoising_aggregator_forward_train_vs_oracle(dev, kw, n, h, w):
    """DenoisingAggregator.forward_train: output, d(<out, R>)/d(input) and d/d(every parameter) -- conv1/2/3, the dense blocks, the fusion's
    conv1 / offset_conv / conv_offset / deformable weight + bias / emb_conv / conv2 -- against autograd over oracle/dcn.py."""
    from oracle.dcn import denoising_aggregator_forward
    m = _conditioned_aggregator(**kw)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, kw['in_channels'], h, w, generator=g)
    R = torch.randn(n, kw['in_channels'], h, w, generator=g)
    sd = {k: v.detach().clone().float() for k, v in m.state_dict().items()}
    names = [k for k, _ in m.named_parameters()]
    for k in names:
        sd[k].requires_grad_(True)
    xr = x.clone().requires_grad_()
    ref = denoising_aggregator_forward(sd, xr, num_blocks=kw.get('num_blocks', 1), num_layers=kw.get('num_layers', 6), emb_nums=kw.get('emb_nums', 3),
                                       with_rdb=kw.get('with_rdb', True), with_taf=kw.get('with_taf', True))
    (ref * R).sum().backward()
    m = m.to(dev).train()
    xd = x.to(dev).requires_grad_()
    out = m(xd)                                   # requires_grad input: forward dispatches to forward_train
    assert out.requires_grad
    err = (out.detach().cpu() - ref.detach()).abs().max().item() / ref.abs().max().item()
    print(f'DenoisingAggregator forward_train output: {err:.2e} of its scale')
    assert err < 2e-2, err
    (out * R.to(dev)).sum().backward()
    params = dict(m.named_parameters())
    rows = [('input',) + grad_metrics(xd.grad, xr.grad)]
    for k in names:
        assert params[k].grad is not None, k
        rows.append((k,) + grad_metrics(params[k].grad, sd[k].grad))
    n_rows = len(rows)
    top = max(r[4] for r in rows)
    # bounds from the first run (fp16 activations and activation gradients): relative L2 <= 4 %, cosine >= 0.9988, single elements up to
    # 19 % of their tensor's largest
    _report('DenoisingAggregator', rows, 2.5e-1, 0.998, 7e-2, top)
    assert n_rows == len(names) + 1

    # how much of that is rounding: the ORACLE with fp16-stored activations / gradients against itself in fp32
    def run_oracle(round_weights):
        sd2 = {k: v.detach().clone().float().cpu() for k, v in m.state_dict().items()}
        if round_weights:
            sd2 = {k: (v.half().float() if v.dim() == 4 else v) for k, v in sd2.items()}
        for k in names:
            sd2[k].requires_grad_(True)
        o = denoising_aggregator_forward(sd2, x, num_blocks=kw.get('num_blocks', 1), num_layers=kw.get('num_layers', 6), emb_nums=kw.get('emb_nums', 3),
                                         with_rdb=kw.get('with_rdb', True), with_taf=kw.get('with_taf', True))
        (o * R).sum().backward()
        return {k: sd2[k].grad for k in names}
    floor = _rounding_floor(run_oracle, names)
    dev_l2 = {k: l2 for k, e, c, l2, s in rows}
    worst = sorted(((dev_l2[k], floor[k][2], k) for k in dev_l2 if k in floor), reverse=True)[:5]
    print('DenoisingAggregator relative L2: device vs fp32 oracle | fp16-rounded oracle vs fp32 oracle (the rounding floor):',
          [(k, round(a, 4), round(b, 4)) for a, b, k in worst])
    med_dev = sorted(dev_l2[k] for k in dev_l2 if k in floor)[len(floor) // 2]
    med_floor = sorted(floor[k][2] for k in dev_l2 if k in floor)[len(floor) // 2]
    print(f'median relative L2: device {med_dev:.4f}, rounding floor {med_floor:.4f}')
    assert med_dev < 2.0 * med_floor + 5e-3, (med_dev, med_floor)


def test_denoising2_aggregator_forward_train_vs_oracle(dev):
    """Denoising2Aggergator.forward_train over three stages (dense blocks only / both / fusion only; a stride-2 hand-over): both returned
    tuples and every gradient (all stage inputs, all_x[-1], every parameter) against autograd over the oracle."""
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    from oracle.dcn import denoising2_aggregator_forward
    from test_dcn import D2_CFG
    m = randomize(registry.build_aggregator(dict(D2_CFG)), seed=21)
    with torch.no_grad():
        gw = torch.Generator().manual_seed(23)
        for l in ('layer2', 'layer3'):
            dp = m.layers[l + '_taf'].dcn_pack
            dp.conv_offset.weight.mul_(0.25)
            dp.weight.copy_(torch.randn(dp.weight.shape, generator=gw) * (2.0 / (9 * dp.weight.shape[1])) ** 0.5)
            dp.bias.copy_(torch.randn(dp.bias.shape, generator=gw) * 0.05)
            for e in m.layers[l + '_taf'].emb_conv:
                e.weight.mul_(0.1)
        for k, p_ in m.named_parameters():
            if k.endswith('conv2.weight') and 'taf' not in k:
                p_.mul_(0.3)
    g = torch.Generator().manual_seed(22)
    n = 3
    x_noise = [torch.randn(n, 32, 24, 32, generator=g), torch.randn(n, 64, 12, 16, generator=g), torch.randn(n, 64, 12, 16, generator=g)]
    all_x = [torch.randn(n, 32, 12, 16, generator=g), torch.randn(n, 32, 12, 16, generator=g)]
    Rn = [torch.randn(t.shape, generator=g) for t in x_noise]
    Ra = torch.randn(all_x[-1].shape, generator=g)
    sd = {k: v.detach().clone().float() for k, v in m.state_dict().items()}
    names = [k for k, _ in m.named_parameters()]
    for k in names:
        sd[k].requires_grad_(True)
    xr = [t.clone().requires_grad_() for t in x_noise]
    ar = [all_x[0], all_x[1].clone().requires_grad_()]
    ref_noise, ref_all = denoising2_aggregator_forward(sd, tuple(xr), tuple(ar), D2_CFG['in_channel'], D2_CFG['layer_name'], D2_CFG['rdb_blocks'],
                                                       D2_CFG['taf_embs'], D2_CFG['downsample'], D2_CFG['with_rdb'], D2_CFG['with_taf'])
    (sum((o * r).sum() for o, r in zip(ref_noise, Rn)) + (ref_all[0] * Ra).sum()).backward()
    m = m.to(dev).train()
    xd = [t.to(dev).requires_grad_() for t in x_noise]
    ad = [all_x[0].to(dev), all_x[1].to(dev).requires_grad_()]
    out_noise, out_all = m(tuple(xd), tuple(ad))
    assert len(out_noise) == 3 and len(out_all) == 2
    for o, r in zip(tuple(out_noise) + (out_all[0],), tuple(ref_noise) + (ref_all[0],)):
        assert (o.detach().cpu() - r.detach()).abs().max().item() < 2e-2 * r.abs().max().item()
    (sum((o * r.to(dev)).sum() for o, r in zip(out_noise, Rn)) + (out_all[0] * Ra.to(dev)).sum()).backward()
    params = dict(m.named_parameters())
    rows = [(f'x_noise[{i}]',) + grad_metrics(xd[i].grad, xr[i].grad) for i in range(3)] + [('all_x[-1]',) + grad_metrics(ad[1].grad, ar[1].grad)]
    for k in names:
        assert params[k].grad is not None, k
        rows.append((k,) + grad_metrics(params[k].grad, sd[k].grad))
    _report('Denoising2Aggergator', rows, 2.5e-1, 0.993, 1.5e-1, max(r[4] for r in rows))          # first run: L2 <= 13.5 %, cosine >= 0.9940 (biases of the dense layers)


def _backbone_grads(dev, m, sd_ref_fn, x, tag, frozen_prefixes, e_big, cos_big, l2_big):
    """Shared body: oracle autograd over sd_ref_fn(sd, x) -> tuple of outputs; device m.forward_train(x); compare all gradients."""
    sd = {k: v.detach().clone().float() for k, v in m.state_dict().items()}
    names = [k for k, p in m.named_parameters() if p.requires_grad]
    for k in names:
        sd[k].requires_grad_(True)
    ref = sd_ref_fn(sd, x)
    g = torch.Generator().manual_seed(77)
    Rs = [torch.randn(r.shape, generator=g) / r.detach().abs().max() for r in ref]
    sum((r * R).sum() for r, R in zip(ref, Rs)).backward()
    m = m.to(dev).train()
    out = m.forward_train(x.to(dev))
    assert len(out) == len(ref)
    for o, r in zip(out, ref):
        assert o.shape == r.shape
        err = (o.detach().cpu() - r.detach()).abs().max().item() / r.detach().abs().max().item()
        assert err < 2e-2, (tag, err)
    sum((o * R.to(dev)).sum() for o, R in zip(out, Rs)).backward()
    params = dict(m.named_parameters())
    rows = []
    top = max(sd[k].grad.abs().max().item() for k in names if sd[k].grad is not None)
    for k, p in params.items():
        if k.startswith(frozen_prefixes) or k not in names:
            assert p.grad is None, k
            continue
        if sd[k].grad is None:
            assert p.grad is None or p.grad.abs().max().item() == 0, k
            continue
        assert p.grad is not None, k
        if sd[k].grad.abs().max().item() < 1e-7 * top:
            continue
        rows.append((k,) + grad_metrics(p.grad, sd[k].grad))
    _report(tag, rows, e_big, cos_big, l2_big, top)
    return rows


def test_insert_resnet_forward_train_vs_oracle(dev):
    """InsertResNet.forward_train: ResNet-50 stages 1-3 with the denoising plugin (dense blocks + DCN temporal fusion) after layer2, a
    3-frame clip; frozen_stages=1 (stem + layer1 get no gradient). Outputs and every trainable gradient -- Bottleneck convs with their
    eval-mode BatchNorm affines, the plugin's convs and deformable conv -- against autograd over the oracle."""
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    from oracle.dcn import denoising_aggregator_forward
    from oracle.detector import resnet_forward
    cfg = dict(num_blocks=1, num_layers=2, emb_nums=1, with_rdb=True, with_taf=True)
    m = randomize(registry.build_backbone(dict(type='InsertResNet', depth=50, num_stages=3, strides=(1, 2, 2), dilations=(1, 1, 1), out_indices=(1, 2),
                                               frozen_stages=1, insert_tag=[False, True, False], insert_cfg=dict(type='DenoisingAggregator', **cfg))),
                  seed=12)
    with torch.no_grad():
        gw = torch.Generator().manual_seed(14)
        for name, mod in m.named_modules():
            if name.endswith('dcn_pack'):
                mod.conv_offset.weight.mul_(0.25)
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=gw) * (2.0 / (9 * mod.weight.shape[1])) ** 0.5)
            if name.endswith('taf'):
                for e in mod.emb_conv:
                    e.weight.mul_(0.3)
    m._freeze_stages()
    x = torch.randn(3, 3, 64, 96, generator=torch.Generator().manual_seed(13))

    def ref_fn(sd, xx):
        def plugin(i, nb, t):
            return denoising_aggregator_forward(sd, t, prefix=f'layer{i + 1}.{nb}.', **cfg) if i == 1 else t
        return resnet_forward(sd, xx, depth=50, out_indices=(1, 2), prefix='', after_stage=plugin, num_stages=3, strides=(1, 2, 2))
    # first run: the offset branch of the plugin's deformable conv inside a 30-conv backbone: cosine >= 0.9931, L2 <= 11.8 %, elements <= 20 %
    rows = _backbone_grads(dev, m, ref_fn, x, 'InsertResNet', ('conv1.', 'bn1.', 'layer1.'), 3e-1, 0.99, 1.6e-1)
    assert any(k.startswith('layer2.4.taf.dcn_pack.') for k, *_ in rows) and any(k.startswith('layer3.5.') for k, *_ in rows)
    assert len(rows) > 100


def test_dark_resnet_forward_train_vs_oracle(dev):
    """DarkResNet.forward_train: the ConvLSTM DenoisingBlock stage (ResNet-50: layer2; style='caffe' so that the stride sits on conv1 and a
    4-frame clip runs) with llv_convlstm_gate / _bwd on the graph: outputs and every gradient, the unfrozen stem (7x7 / stride-2 conv,
    max pooling backward) included, against autograd over the oracle's recurrence."""
    from lowlightenvironmentvideoobjectdetection_amd.registry import build_backbone
    from helpers import smooth_image
    from oracle.detector import resnet_forward
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    kw = dict(style='caffe', strides=(1, 2), dilations=(1, 1), out_indices=(0, 1), num_stages=2)
    m = randomize(build_backbone(dict(type='DarkResNet', depth=50, seqs=4, frozen_stages=-1, **kw)), seed=55)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.startswith('layer2.') and n_.endswith('conv2.weight'):
                p.mul_(3.0)
    x = smooth_image(4, 48, 64, seed=7)

    def ref_fn(sd, xx):
        return resnet_forward(sd, xx, depth=50, prefix='', dark_stages=(1,), strides=kw['strides'], num_stages=2, style='caffe', out_indices=(0, 1))
    rows = _backbone_grads(dev, m, ref_fn, x, 'DarkResNet', ('\x00',), 1.5e-1, 0.997, 8e-2)       # first run: cosine >= 0.9985, L2 <= 5.4 %, elements <= 9.8 %
    assert any(k == 'conv1.weight' for k, *_ in rows) and any(k == 'layer2.3.conv2.weight' for k, *_ in rows) and any(k == 'layer2.0.bn2.weight' for k, *_ in rows)


def test_authors_config_aggregator_trains(dev):
    """The aggregator dict of the authors' config mmtracking/configs/vid/llvod/llvod_l1234_fusion_add_i1234_rdb_taf_darkfarm.py:146-159
    (Denoising2Aggergator over the four ResNet-50 stages, 256..2048 channels, dense blocks + DCN temporal fusion at every stage), unchanged,
    through forward_train on a 2-frame clip of small maps: the training path's outputs equal the inference plan's (same kernels, other
    launch structure) and one backward pass leaves a finite, non-zero gradient on every one of its parameters and on every input."""
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    cfg = dict(type='Denoising2Aggergator', in_channel=[256, 512, 1024, 2048], mid_channel=[64, 128, 256, 512], out_channel=[512, 1024, 2048, 512],
               layer_name=['layer1', 'layer2', 'layer3', 'layer4'], rdb_blocks=[2, 2, 4, 2], rdb_channel_growth=[64, 64, 128, 128],
               taf_embs=[3, 3, 3, 3], downsample=[True, True, False, False], with_rdb=[True, True, True, True], with_taf=[True, True, True, True])
    m = randomize(registry.build_aggregator(cfg), seed=31)
    with torch.no_grad():
        gw = torch.Generator().manual_seed(32)
        for name, mod in m.named_modules():
            if name.endswith('dcn_pack'):
                mod.conv_offset.weight.mul_(0.25)
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=gw) * (2.0 / (9 * mod.weight.shape[1])) ** 0.5)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=gw) * 0.05)
            if name.endswith('_taf'):
                for e in mod.emb_conv:
                    e.weight.mul_(0.1)
        for k, p_ in m.named_parameters():
            if k.endswith('conv2.weight') and 'taf' not in k:
                p_.mul_(0.3)
    m = m.to(dev).train()
    g = torch.Generator().manual_seed(33)
    n = 2
    x_noise = [torch.randn(n, 256, 16, 24, generator=g), torch.randn(n, 512, 8, 12, generator=g), torch.randn(n, 1024, 4, 6, generator=g),
               torch.randn(n, 2048, 4, 6, generator=g)]
    all_x = [torch.randn(n, 512, 4, 6, generator=g)]
    with torch.no_grad():
        plan_noise, plan_all = m(tuple(t.to(dev) for t in x_noise), tuple(t.to(dev) for t in all_x))
        plan_noise, plan_all = [t.clone() for t in plan_noise], [t.clone() for t in plan_all]
    xd = [t.to(dev).requires_grad_() for t in x_noise]
    ad = [all_x[0].to(dev).requires_grad_()]
    out_noise, out_all = m.forward_train(tuple(xd), tuple(ad))
    for o, r in zip(tuple(out_noise) + (out_all[0],), tuple(plan_noise) + (plan_all[0],)):
        assert o.shape == r.shape and (o.detach() - r).abs().max().item() < 2e-2 * r.abs().max().item()
    R = [torch.randn(o.shape, generator=g).to(dev) for o in tuple(out_noise) + (out_all[0],)]
    sum((o * r).sum() for o, r in zip(tuple(out_noise) + (out_all[0],), R)).backward()
    bad = [k for k, p in m.named_parameters() if p.grad is None or not bool(torch.isfinite(p.grad).all())]
    assert bad == [], bad
    # a bias in front of the softmax over the frames has an exactly zero gradient (test above); every other tensor gets a real one
    zero_ok = ('dcn_pack.bias', 'emb_conv.0.bias', 'emb_conv.1.bias', 'emb_conv.2.bias')
    dead = [k for k, p in m.named_parameters() if p.grad.abs().max().item() == 0 and not k.endswith(zero_ok)]
    assert dead == [], dead
    assert all(t.grad is not None and bool(torch.isfinite(t.grad).all()) and t.grad.abs().max().item() > 0 for t in xd + ad)
    assert len(list(m.parameters())) > 150
