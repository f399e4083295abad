"""The implicit-GEMM tiles on SINGLE-IMAGE maps (the training step and the per-frame protocol: 3,680- and 920-pixel maps fill a fraction of
the chip): microseconds per launch inside a captured graph of 20 identical launches, per forced tile and for the planner's own choice.
python tools/sweep_single_image.py"""
import sys

import torch

sys.path.insert(0, '.')
from lowlightenvironmentvideoobjectdetection_amd import ops
from lowlightenvironmentvideoobjectdetection_amd._lib import CONV_RELU, CONV_RES_ADD

dev = torch.device('cuda:0')
TILES = {0: 'auto', 1: '128x128', 5: '64x64', 14: '64x64K64', 7: '128x128K64'}
shapes = [(92, 160, 512, 128, 1, False, 'layer2 conv1'), (92, 160, 128, 512, 1, True, 'layer2 conv3'), (92, 160, 128, 128, 3, False, 'layer2 conv2'),
          (46, 80, 1024, 256, 1, False, 'layer3 conv1'), (46, 80, 256, 1024, 1, True, 'layer3 conv3'), (46, 80, 256, 256, 3, False, 'layer3 conv2'),
          (23, 40, 2048, 512, 1, False, 'layer4 conv1'), (23, 40, 512, 2048, 1, True, 'layer4 conv3'), (23, 40, 512, 512, 3, False, 'layer4 conv2'),
          (92, 160, 256, 256, 3, False, 'fpn P3 conv'), (46, 80, 256, 256, 3, False, 'fpn P4 conv')]
for hh, ww, k, n, ks, res, name in shapes:
    g = torch.Generator().manual_seed(1)
    wt = torch.randn(n, k, ks, ks, generator=g) / (k * ks * ks) ** 0.5
    pc = ops.pack_conv(wt, torch.randn(n, generator=g), stride=1, pad=ks // 2).to(dev)
    x = torch.randn(1, hh, ww, k, device=dev).half()
    ident = torch.randn(1, hh, ww, n, device=dev).half() if res else None
    y = torch.empty(1, hh, ww, n, device=dev, dtype=torch.float16)
    flags = CONV_RELU | (CONV_RES_ADD if res else 0)
    line = []
    for tile, tn in TILES.items():
        kw = dict(tile=tile) if tile else {}
        try:
            ops.conv2d(x, pc, res=ident, flags=flags, out=y, **kw)
            torch.cuda.synchronize()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                ops.conv2d(x, pc, res=ident, flags=flags, out=y, **kw)
            torch.cuda.current_stream().wait_stream(s)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(20):
                    ops.conv2d(x, pc, res=ident, flags=flags, out=y, **kw)
        except Exception as e:          # a tile that does not take this shape
            line.append(f'{tn}: n/a')
            continue
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        line.append(f'{tn}: {e0.elapsed_time(e1) * 1e3 / 100:5.1f}')
    fl = 2.0 * hh * ww * k * n * ks * ks
    print(f'{name:13s} {hh * ww:6d} px {k:4d}->{n:4d} k{ks} {fl * 1e-9:5.1f} GF | ' + ' | '.join(line), flush=True)
