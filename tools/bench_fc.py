"""Split-K choice for the small-M fully connected layers of the SELSA RoI head (M = 300 proposals): fc0 = 7x7-valid conv over 256 channels
(K = 12544 -> 1024), fc1 = 1x1 (1024 -> 1024).  Usage: python tools/bench_fc.py"""
import sys, torch
sys.path.insert(0, '.')
from lowlightenvironmentvideoobjectdetection_amd import ops
dev = torch.device('cuda:0')
for name, cin, k, hw in (('fc0 7x7-valid 256->1024', 256, 7, 7), ('fc 1x1 1024->1024', 1024, 1, 1)):
    g = torch.Generator().manual_seed(1)
    wt = torch.randn(1024, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    pc = ops.pack_conv(wt, torch.randn(1024, generator=g), stride=1, pad=0).to(dev)
    x = torch.randn(300, hw, hw, cin, device=dev).half()
    row = []
    ref = None
    for tile, splits in ((0, 0), (5, 4), (5, 7), (5, 14), (5, 28), (5, 49), (1, 8), (1, 16), (1, 32)):
        try:
            y = ops.conv2d(x, pc, flags=ops.CONV_RELU, tile=tile, k_splits=splits)
        except Exception as e:
            row.append(f'{tile}/{splits}: n/a'); continue
        torch.cuda.synchronize()
        if ref is None:
            ref = y.float()
        err = (y.float() - ref).abs().max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.conv2d(x, pc, flags=ops.CONV_RELU, tile=tile, k_splits=splits, out=y)
        e1.record(); torch.cuda.synchronize()
        row.append(f'{tile}/{splits}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} (d={err:.2g})')
    print(f'{name:26s} ' + '  '.join(row))
