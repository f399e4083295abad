"""llv_conv2d_nhwc_wgrad alone on the shapes of the training step (knock-outs: LLV_WGRAD_KNOCK with a library built by `make EXTRA=-DLLV_KNOCKOUTS`) (one 736 x 1280 image, RetinaNet R-50 + FPN): microseconds and
TFLOP/s per launch, nothing else on the device.  python tools/wgrad_bench.py [name-filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lowlightenvironmentvideoobjectdetection_amd import backward as B

SHAPES = [  # name, h, w, cin, cout, groups, k, stride
    ('head.tower P3 (2 groups)', 92, 160, 256, 256, 2, 3, 1),
    ('head.tower P4 (2 groups)', 46, 80, 256, 256, 2, 3, 1),
    ('head.tower P5 (2 groups)', 23, 40, 256, 256, 2, 3, 1),
    ('head.cls P3', 92, 160, 256, 720, 1, 3, 1),
    ('head.reg P3', 92, 160, 256, 36, 1, 3, 1),
    ('fpn.conv P3', 92, 160, 256, 256, 1, 3, 1),
    ('fpn.lateral C3', 92, 160, 512, 256, 1, 1, 1),
    ('layer2.conv1', 92, 160, 512, 128, 1, 1, 1),
    ('layer2.conv2', 92, 160, 128, 128, 1, 3, 1),
    ('layer2.conv3', 92, 160, 128, 512, 1, 1, 1),
    ('layer3.conv1', 46, 80, 1024, 256, 1, 1, 1),
    ('layer3.conv2', 46, 80, 256, 256, 1, 3, 1),
    ('layer3.conv3', 46, 80, 256, 1024, 1, 1, 1),
    ('layer3.0.conv2 s2', 92, 160, 256, 256, 1, 3, 2),
    ('layer4.conv1', 23, 40, 2048, 512, 1, 1, 1),
    ('layer4.conv2', 23, 40, 512, 512, 1, 3, 1),
    ('layer4.conv3', 23, 40, 512, 2048, 1, 1, 1),
]


def main():
    dev = torch.device('cuda:0')
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
    tot = 0.0
    for name, h, w, cin, cout, g, k, s in SHAPES:
        if flt not in name:
            continue
        pad = k // 2
        ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        x = torch.randn(h * w, g * cin, device=dev).half()
        dy_ld = (g * cout + 7) // 8 * 8
        dy = torch.randn(ho * wo, dy_ld, device=dev).half()
        w_rows, k_pad = (cout + 7) // 8 * 8, k * k * cin
        dw = torch.zeros(g, w_rows, k_pad, device=dev)
        run = lambda: B.conv_wgrad(x, dy, dw, cin_pad=cin, cout=cout, w_rows=w_rows, k_pad=k_pad, kh=k, kw=k, stride=s, pad=pad, groups=g,
                                   batch=1, seg=(h, w, ho, wo, 0, 0), x_ld=g * cin, dy_ld=dy_ld)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        fl = 2.0 * ho * wo * cin * cout * g * k * k
        tot += us
        print(f'{name:28s} {ho * wo:6d} px  {cin:4d}->{cout:4d} x{g} k{k} s{s}  {us:7.1f} us  {fl / us * 1e-6:7.1f} TFLOP/s  dW {dw.numel() * 4 / 1e6:5.1f} MB')
    print(f'sum {tot:.1f} us')


if __name__ == '__main__':
    main()
