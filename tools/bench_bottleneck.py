"""Times llv_bottleneck_fwd against the three conv launches it replaces at the detector's full-size shapes (HIP events, graph-free).
Usage (GPU box): python tools/bench_bottleneck.py"""
import sys
import torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from lowlightenvironmentvideoobjectdetection_amd import engine, ops
from lowlightenvironmentvideoobjectdetection_amd._lib import CONV_RELU, CONV_RES_ADD
from test_gpu_bottleneck import _block_params, _bn

dev = torch.device('cuda:0')
for mid, n, h, w in ((64, 16, 184, 320), (128, 16, 92, 160)):
    sd = _block_params(mid, 1)
    pcs = lambda: (ops.pack_conv(sd['conv1.weight'], None, 1, 0, bn=_bn(sd, 1)), ops.pack_conv(sd['conv2.weight'], None, 1, 1, bn=_bn(sd, 2)),
                   ops.pack_conv(sd['conv3.weight'], None, 1, 0, bn=_bn(sd, 3)))
    b1 = engine.Builder(dev)
    xin = b1.buf(n, h, w, 4 * mid)
    xin.copy_(torch.randn(n, h, w, 4 * mid, device=dev).relu().half())
    p1, p2, p3 = pcs()
    o = b1.conv('conv1', xin, p1, CONV_RELU)
    o = b1.conv('conv2', o, p2, CONV_RELU)
    y = b1.conv('conv3', o, p3, CONV_RELU | CONV_RES_ADD, res=xin)
    b2 = engine.Builder(dev)
    q = pcs()
    y2 = b2.bottleneck('fused', xin, *q)
    for b in (b1, b2):
        b.plan.run()
    torch.cuda.synchronize()
    print(f'mid={mid} {n}x{h}x{w}: max |fused - separate| = {(y.float() - y2.float()).abs().max().item():.4g} (scale {y.float().abs().max().item():.3g})')
    for label, b in (('separate', b1), ('fused', b2)):
        rows = b.plan.run_timed(iters=10)
        tot = sum(r[1] for r in rows)
        fl = sum(r[2] for r in rows)
        print(f'  {label:9s} {tot * 1e3:8.1f} us  {fl / tot / 1e9:7.1f} TFLOP/s   ' + '  '.join(f'{r[0]}={r[1] * 1e3:.1f}us' for r in rows))
