"""Which implicit-GEMM tile is fastest for each 1x1 convolution of the backbone at the 16-frame step's shapes? conv1 = bias + ReLU,
conv3 = bias + identity + ReLU. Between timed launches a 512 MB buffer is rewritten so that the operands come from HBM, not from the
Infinity Cache (as in the step, where other launches run in between).  python tools/sweep_1x1_tiles.py"""
import sys

import torch

sys.path.insert(0, '.')
from lowlightenvironmentvideoobjectdetection_amd import ops
from lowlightenvironmentvideoobjectdetection_amd._lib import CONV_RELU, CONV_RES_ADD

dev = torch.device('cuda:0')
TILES = {1: '128x128', 2: '64x256', 3: '32x256', 5: '64x64', 7: '128x128K64', 8: '64x256K64'}
shapes = [(46, 80, 1024, 256, False, 'layer3 conv1'), (46, 80, 256, 1024, True, 'layer3 conv3'), (23, 40, 2048, 512, False, 'layer4 conv1'),
          (23, 40, 512, 2048, True, 'layer4 conv3'), (92, 160, 512, 128, False, 'layer2.0 conv1'), (184, 320, 256, 64, False, 'layer1 conv1')]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for hh, ww, k, n, res, name in shapes:
    m = 16 * hh * ww
    g = torch.Generator().manual_seed(1)
    wt = torch.randn(n, k, 1, 1, generator=g) / k ** 0.5
    pc = ops.pack_conv(wt, torch.randn(n, generator=g), stride=1, pad=0).to(dev)
    x = torch.randn(16, hh, ww, k, device=dev).half()
    ident = torch.randn(16, hh, ww, n, device=dev).half() if res else None
    y = torch.empty(16, hh, ww, n, device=dev, dtype=torch.float16)
    flags = CONV_RELU | (CONV_RES_ADD if res else 0)
    line = []
    for tile, tn in TILES.items():
        try:
            ops.conv2d(x, pc, res=ident, flags=flags, out=y, tile=tile)
        except Exception as e:          # a tile that does not take this shape
            line.append(f'{tn}: n/a')
            continue
        torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            flush.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv2d(x, pc, res=ident, flags=flags, out=y, tile=tile)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        line.append(f'{tn}: {ts[len(ts) // 2]:6.1f}')
    auto = []
    for _ in range(8):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv2d(x, pc, res=ident, flags=flags, out=y)
        e1.record()
        torch.cuda.synchronize()
        auto.append(e0.elapsed_time(e1) * 1e3)
    auto.sort()
    print(f'{name:15s} M={m:7d} K={k:5d} N={n:5d} | auto {auto[len(auto) // 2]:6.1f} us | ' + ' | '.join(line), flush=True)
