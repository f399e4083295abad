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
    rows.sort(key=lambda t: -t[1])
    print(tag, 'largest max-abs errors:', [(k, round(e, 3), round(c, 5), round(l2, 3)) for k, e, c, l2, s in rows[:6]])
    print(tag, 'lowest cosines:', [(k, round(c, 5)) for k, e, c, l2, s in sorted(rows, key=lambda t: t[2])[:4]])
    for k, e, c, l2, s in rows:
        big = s > 1e-3 * top
        assert e < (e_big if big else 3 * e_big), (k, e, s)
        assert c > (cos_big if big else 1 - 10 * (1 - cos_big)), (k, c, s)
        assert l2 < (l2_big if big else 3 * l2_big), (k, l2, s)


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
    assert err < 2e-2, err
    (out * R.to(dev)).sum().backward()
    params = dict(m.named_parameters())
    rows = [('input',) + grad_metrics(xd.grad, xr.grad)]
    for k in names:
        assert params[k].grad is not None, k
        rows.append((k,) + grad_metrics(params[k].grad, sd[k].grad))
    _report('DenoisingAggregator', rows, 6e-2, 0.998, 5e-2, max(r[4] for r in rows))
    assert len(rows) == len(names) + 1


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
    _report('Denoising2Aggergator', rows, 8e-2, 0.997, 6e-2, max(r[4] for r in rows))


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
    rows = _backbone_grads(dev, m, ref_fn, x, 'InsertResNet', ('conv1.', 'bn1.', 'layer1.'), 8e-2, 0.997, 6e-2)
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
    rows = _backbone_grads(dev, m, ref_fn, x, 'DarkResNet', ('\x00',), 8e-2, 0.997, 6e-2)
    assert any(k == 'conv1.weight' for k, *_ in rows) and any(k == 'layer2.3.conv2.weight' for k, *_ in rows) and any(k == 'layer2.0.bn2.weight' for k, *_ in rows)
