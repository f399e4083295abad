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
    # stage breakdown from the diagnostic launch (s_memtime stamps per workgroup)
    import ctypes as C
    from lowlightenvironmentvideoobjectdetection_amd._lib import lib, BottleneckDesc
    fn = lib().llv_dbg_bottleneck_stamps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    d = [s_[2][0]._obj for s_ in b2.plan.steps][0]
    import os
    tw = 32 if (mid == 64 and os.environ.get('LLV_BNK_SMALL', '1') == '0') else 16
    nblk = n * ((h + 7) // 8) * ((w + tw - 1) // tw)
    st = torch.zeros(nblk, 4, dtype=torch.int64, device=dev)
    rc = fn(C.addressof(d), st.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0
    t = st.cpu().double()
    t0 = t[:, 0].min()
    dur = (t[:, 3] - t[:, 0])
    print(f'  stamps (ticks): kernel span {t[:, 3].max() - t0:.0f}; per workgroup: total mean {dur.mean():.0f} (min {dur.min():.0f} max {dur.max():.0f}); '
          f'stage1 {(t[:, 1] - t[:, 0]).mean():.0f}  stage2 {(t[:, 2] - t[:, 1]).mean():.0f}  stage3 {(t[:, 3] - t[:, 2]).mean():.0f}; '
          f'workgroups {nblk} = {nblk / 256:.1f} per CU; sum of workgroup time / 256 = {dur.sum() / 256:.0f}')
