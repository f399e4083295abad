"""Tile / split-K choice for single-image launches (M = 3680 pixels): auto plan against forced 128x128 tiles with split-K.
Usage: python tools/bench_small_m.py"""
import sys, torch
sys.path.insert(0, '.')
from lowlightenvironmentvideoobjectdetection_amd import ops
dev = torch.device('cuda:0')
cases = [(512, 512, 3, 2, 'layer4 conv2 dilated'), (256, 256, 3, 1, 'layer3 conv2'), (1024, 256, 1, 1, 'layer3 conv1'), (256, 1024, 1, 1, 'layer3 conv3'),
         (2048, 512, 1, 1, 'layer4 conv1 (DC5)'), (512, 2048, 1, 1, 'layer4 conv3 (DC5)')]
for cin, cout, k, dil, name in cases:
    g = torch.Generator().manual_seed(1)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    pc = ops.pack_conv(wt, torch.randn(cout, generator=g), stride=1, pad=dil * (k // 2), dilation=dil).to(dev)
    x = torch.randn(1, 46, 80, cin, device=dev).half()
    row = []
    for tile, splits in ((0, 0), (1, 1), (1, 2), (1, 4), (1, 8), (5, 1), (5, 2), (5, 4)):
        try:
            y = ops.conv2d(x, pc, flags=ops.CONV_RELU, tile=tile, k_splits=splits)
        except Exception as e:
            row.append(f'{tile}/{splits}: n/a'); continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.conv2d(x, pc, flags=ops.CONV_RELU, tile=tile, k_splits=splits, out=y)
        e1.record(); torch.cuda.synchronize()
        row.append(f'{tile}/{splits}: {e0.elapsed_time(e1) / 30 * 1e3:.1f}')
    print(f'{name:24s} ' + '  '.join(row))
