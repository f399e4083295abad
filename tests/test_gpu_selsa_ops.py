"""The SELSA RoI head's matrix products on the in-tree kernels: fully connected layers as 1x1 / 7x7-valid
convolutions, the proposal attention as two grouped convolutions around llv_selsa_softmax, against torch fp32 on the CPU (the arithmetic
selsa_aggregator.py:44-71 / F.linear perform), and the small helpers of the fixed-shape step."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize('n,m,c,nb', [(37, 70, 128, 4), (300, 900, 1024, 16), (5, 3, 64, 8)])
def test_selsa_attention_vs_torch(dev, n, m, c, nb):
    from lowlightenvironmentvideoobjectdetection_amd.selsa import selsa_attention
    g = torch.Generator().manual_seed(n + m)
    xe, re, rn = (torch.randn(n, c, generator=g).half(), torch.randn(m, c, generator=g).half(), torch.randn(m, c, generator=g).half())
    d = c // nb
    q = xe.float().view(n, nb, d).permute(1, 0, 2)
    k = re.float().view(m, nb, d).permute(1, 2, 0)
    w = (torch.bmm(q, k) / d ** 0.5).softmax(dim=2)
    want = torch.bmm(w, rn.float().view(m, nb, d).permute(1, 0, 2)).permute(1, 0, 2).reshape(n, c)
    got = selsa_attention(xe.to(dev), re.to(dev), rn.to(dev), nb).float().cpu()
    assert (got - want).abs().max().item() < 4e-3 * want.abs().max().item() + 2e-3


def test_linear_layers_on_the_conv_engine(dev):
    import torch.nn as nn
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from lowlightenvironmentvideoobjectdetection_amd._lib import CONV_OUT_F32
    from lowlightenvironmentvideoobjectdetection_amd.selsa import _lin, linear_pc
    torch.manual_seed(0)
    fc = nn.Linear(512, 1000).to(dev)
    x = torch.randn(77, 512, device=dev)
    want = torch.nn.functional.linear(x.half().float().cpu(), fc.weight.half().float().cpu(), fc.bias.float().cpu())
    assert (_lin(fc, x).cpu() - want).abs().max().item() < 2e-3 * want.abs().max().item()
    # first shared fc: flattened (C, 7, 7) RoI map == 7x7 valid convolution over the NHWC RoI features
    fc0 = nn.Linear(64 * 49, 256).to(dev)
    roi = torch.randn(33, 64, 7, 7, device=dev)
    want = torch.nn.functional.linear(roi.flatten(1).half().float().cpu(), fc0.weight.half().float().cpu(), fc0.bias.float().cpu())
    y = ops.conv2d(roi.permute(0, 2, 3, 1).contiguous().half(), linear_pc(fc0, roi_hw=(7, 7)), flags=CONV_OUT_F32)
    assert y.shape == (33, 1, 1, 256)
    assert (y.view(33, 256).cpu() - want).abs().max().item() < 2e-3 * want.abs().max().item()
    with torch.no_grad():
        fc.bias.add_(1.0)                       # an in-place parameter update re-packs
    assert abs((_lin(fc, x).cpu() - want.new_zeros(1)).mean().item() - (torch.nn.functional.linear(x.cpu(), fc.weight.cpu(), fc.bias.cpu())).mean().item()) < 1e-2


def test_masked_softmax_rois_and_live_count(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from lowlightenvironmentvideoobjectdetection_amd._lib import check, lib
    L = lib()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(10, 36, generator=g) * 3
    cnt = torch.tensor([7], dtype=torch.int32, device=dev)
    y = torch.full((10, 40), -1.0, device=dev)
    check(L.llv_softmax_rows_masked(ops._ptr(x.to(dev)), ops._ptr(y), 10, 31, 36, 40, ops._ptr(cnt), _st()), 'llv_softmax_rows_masked')
    want = x[:, :31].softmax(1)
    assert torch.allclose(y[:7, :31].cpu(), want[:7], atol=1e-6) and float(y[7:, :31].abs().max()) == 0 and float(y[:, 31:].min()) == -1.0
    dets = torch.rand(6, 5, generator=g)
    rois = torch.full((6, 5), 9.0, device=dev)
    b4 = torch.full((6, 4), 9.0, device=dev)
    c2 = torch.tensor([4], dtype=torch.int32, device=dev)
    check(L.llv_dets_to_rois(ops._ptr(dets.to(dev)), ops._ptr(c2), ops._ptr(rois), ops._ptr(b4), 6, 2, _st()), 'llv_dets_to_rois')
    assert torch.equal(rois[:4, 1:].cpu(), dets[:4, :4]) and float(rois[:4, 0].min()) == 2.0 and float(rois[4:].abs().max()) == 0
    assert torch.equal(b4[:4].cpu(), dets[:4, :4]) and float(b4[4:].abs().max()) == 0
    # softmax over the reference rows with a live count: static rows [0, 5) + 3 of 6 live rows
    s = torch.randn(4, 2, 32, generator=g)
    p = torch.zeros(4, 2, 32, dtype=torch.float16, device=dev)
    live = torch.tensor([3], dtype=torch.int32, device=dev)
    check(L.llv_selsa_softmax(ops._ptr(s.to(dev)), ops._ptr(p), 4, 2, 32, 5, 6, ops._ptr(live), 0.5, _st()), 'llv_selsa_softmax')
    want = (s[..., :8] * 0.5).softmax(-1)
    assert (p[..., :8].float().cpu() - want).abs().max().item() < 2e-3 and float(p[..., 8:].abs().max()) == 0
