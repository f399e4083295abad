"""The implicit-GEMM tiles on the stride-2 3x3 convolutions of the 16-frame step (FastDVDnet's DownBlocks, ResNet's layerN.0.conv2): us per
launch (median of 8, a 512 MB buffer rewritten in between so that the operands come from HBM) per forced tile and for the planner's choice.
python tools/sweep_stride2.py"""
import sys

import torch

sys.path.insert(0, '.')
from lowlightenvironmentvideoobjectdetection_amd import ops
from lowlightenvironmentvideoobjectdetection_amd._lib import CONV_RELU

dev = torch.device('cuda:0')
TILES = {0: 'auto', 1: '128x128', 2: '64x256', 3: '32x256', 5: '64x64', 7: '128x128K64', 8: '64x256K64'}
shapes = [(18, 736, 1280, 32, 64, 'temp1.down0.0'), (18, 368, 640, 64, 128, 'temp1.down1.0'), (16, 184, 320, 128, 128, 'layer2.0.conv2'),
          (16, 92, 160, 256, 256, 'layer3.0.conv2'), (16, 46, 80, 512, 512, 'layer4.0.conv2'), (16, 23, 40, 2048, 256, 'neck.fpn3 (P6)')]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for nb, hh, ww, k, n, name in shapes:
    g = torch.Generator().manual_seed(1)
    wt = torch.randn(n, k, 3, 3, generator=g) / (9 * k) ** 0.5
    pc = ops.pack_conv(wt, torch.randn(n, generator=g), stride=2, pad=1).to(dev)
    x = torch.randn(nb, hh, ww, k, device=dev).half()
    ho, wo = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
    y = torch.empty(nb, ho, wo, n, device=dev, dtype=torch.float16)
    line = []
    for tile, tn in TILES.items():
        kw = dict(tile=tile) if tile else {}
        try:
            ops.conv2d(x, pc, flags=CONV_RELU, out=y, **kw)
        except Exception:
            line.append(f'{tn}: n/a')
            continue
        torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            flush.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv2d(x, pc, flags=CONV_RELU, out=y, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        line.append(f'{tn}: {ts[len(ts) // 2]:6.1f}')
    fl = 2.0 * nb * ho * wo * k * n * 9
    print(f'{name:15s} {nb * ho * wo:8d} px {k:4d}->{n:4d} {fl * 1e-9:6.1f} GF | ' + ' | '.join(line), flush=True)
