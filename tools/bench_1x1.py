"""1x1 convolutions of the backbone at the 16-frame step's shapes: conv_igemm_glds (libllv) against the library GEMM torch dispatches
to (hipBLASLt / rocBLAS), fp16 in, fp32 accumulate. Usage: python tools/bench_1x1.py"""
import sys
import torch
sys.path.insert(0, '.')
from lowlightenvironmentvideoobjectdetection_amd import ops

dev = torch.device('cuda:0')
shapes = [(46, 80, 1024, 256, 'layer3 conv1'), (46, 80, 256, 1024, 'layer3 conv3'), (23, 40, 2048, 512, 'layer4 conv1'),
          (23, 40, 512, 2048, 'layer4 conv3'), (92, 160, 512, 128, 'layer2 conv1'), (92, 160, 128, 512, 'layer2 conv3'),
          (184, 320, 64, 256, 'layer1.0 conv3')]
for hh, ww, k, n, name in shapes:
    m = 16 * hh * ww
    g = torch.Generator().manual_seed(1)
    wt = torch.randn(n, k, 1, 1, generator=g) / k ** 0.5
    pc = ops.pack_conv(wt, torch.randn(n, generator=g), stride=1, pad=0).to(dev)
    x = torch.randn(16, hh, ww, k, device=dev).half()
    y = ops.conv2d(x, pc, flags=ops.CONV_RELU)
    a = x.view(m, k)
    w2 = wt.view(n, k).t().contiguous().half().to(dev)
    bias = torch.zeros(n, device=dev, dtype=torch.half)
    def run_llv():
        ops.conv2d(x, pc, flags=ops.CONV_RELU, out=y)
    def run_lib():
        torch.relu_(torch.addmm(bias, a, w2))
    res = []
    for fn in (run_llv, run_lib):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
    fl = 2.0 * m * k * n
    print(f'{name:16s} M={m:7d} K={k:5d} N={n:5d}: libllv {res[0] * 1e3:7.1f} us ({fl / res[0] / 1e9:6.1f} TF/s)   library GEMM + relu {res[1] * 1e3:7.1f} us ({fl / res[1] / 1e9:6.1f} TF/s)')
