"""A/B timing of conv tile variants on representative layer shapes (one process, interleaved rounds)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from lowlightenvironmentvideoobjectdetection_amd import ops
dev = torch.device('cuda:0')
shapes = [  # name, n, cin, cout, h, w, k, stride
    ('fdvd128 3x3 180x320 b6', 6, 128, 128, 184, 320, 3, 1),
    ('head tower-like 256->256 3x3 92x160 b4', 4, 256, 256, 92, 160, 3, 1),
    ('layer2 1x1 512->128 92x160 b4', 4, 512, 128, 92, 160, 1, 1),
    ('layer1 1x1 64->256 184x320 b4', 4, 64, 256, 184, 320, 1, 1),
    ('layer3 3x3 256->256 46x80 b4', 4, 256, 256, 46, 80, 3, 1),
    ('cls 256->720 3x3 92x160 b4', 4, 256, 720, 92, 160, 3, 1),
    ('fdvd64 3x3 368x640 b6', 6, 64, 64, 368, 640, 3, 1),
    ('fdvd up1.2 64->128 3x3 368x640 b6', 6, 64, 128, 368, 640, 3, 1),
    ('fdvd inc1 96->32 3x3 736x1280 b2', 2, 96, 32, 736, 1280, 3, 1),
    ('retina_reg 256->36 3x3 92x160 b1', 1, 256, 36, 92, 160, 3, 1),
    ('fpn out 256->256 3x3 23x40 b1', 1, 256, 256, 23, 40, 3, 1),
]
tiles = [int(t) for t in (sys.argv[1].split(',') if len(sys.argv) > 1 else '1,6'.split(','))]
if len(sys.argv) > 2:                                     # optional: indices of the shapes to run
    shapes = [shapes[int(i)] for i in sys.argv[2].split(',')]
for name, n, cin, cout, h, w, k, st in shapes:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, h, w, cin, generator=g).half().to(dev)
    pc = ops.pack_conv(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, torch.zeros(cout), st, (k - 1) // 2).to(dev)
    ho, wo = ops.conv_out_size(h, w, pc)
    flops = 2.0 * n * ho * wo * cout * cin * k * k
    outs = {t: torch.empty(n, ho, wo, pc.cout, dtype=torch.float16, device=dev) for t in tiles}
    res = {t: [] for t in tiles}
    for rnd in range(6):
        for t in tiles:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.conv2d(x, pc, flags=ops.CONV_RELU, out=outs[t], tile=t, k_splits=0 if t in (9, 10, 11) else 1)
            e1.record(); torch.cuda.synchronize()
            if rnd > 0:
                res[t].append(e0.elapsed_time(e1) / 5)
    ref = outs[tiles[0]].float()
    line = f'{name:42s}'
    for t in tiles:
        ms = sorted(res[t])[len(res[t]) // 2]
        err = (outs[t].float() - ref).abs().max().item()
        line += f' | tile{t}: {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF/s (d={err:.3g})'
    print(line)
