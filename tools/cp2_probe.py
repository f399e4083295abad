"""Phase breakdown of the fused CvBlock kernels (convpair2_kernel<8,32> = InputCvBlock, <32,16> = OutputCvBlock) at full size, from the
s_memtime stamps of the diagnostic library tools/_build/libllv_cp2probe.so (tools/build_cp2_probe.sh; the product library has no
stamps). Also times the launches with HIP events.   Usage (GPU box):  python tools/cp2_probe.py [images]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lowlightenvironmentvideoobjectdetection_amd import _lib  # noqa: E402

probe = os.environ.get('CP2_LIB') or os.path.join(ROOT, 'tools', '_build', 'libllv_cp2probe.so')
use_probe = os.path.exists(probe) and os.environ.get('CP2_NO_STAMPS', '0') != '1'
if use_probe:
    _lib.LIB_PATH = probe
from lowlightenvironmentvideoobjectdetection_amd import engine, ops  # noqa: E402
from lowlightenvironmentvideoobjectdetection_amd._lib import lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
h, w = 736, 1280
dev = torch.device('cuda:0')
torch.manual_seed(0)
import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401,E402
from lowlightenvironmentvideoobjectdetection_amd import registry  # noqa: E402
from lowlightenvironmentvideoobjectdetection_amd.synthetic import seeded_weights  # noqa: E402
cleaner = seeded_weights(registry.build_cleaner(dict(type='FastDVDnet')), seed=1)
plan = engine.FastDVDnetPlan(cleaner.state_dict(), n, h, w, dev)
plan.frames.copy_((torch.randn(plan.frames.shape, device=dev) * 0.5).half())
plan.frames[..., 3:] = 0
plan.plan.run()
torch.cuda.synchronize()
names = [s[0] for s in plan.plan.steps]
rows = plan.plan.run_timed(iters=5)
for r in rows:
    if 'cv' in r[0]:
        print(f'{r[0]:18s} {r[1] * 1e3:9.1f} us  {r[2] / max(r[1], 1e-9) / 1e9:8.1f} TFLOP/s  ({plan.plan.kernel_of.get(r[0])})')
tot = sum(r[1] for r in rows)
print(f'FastDVDnet plan ({n} frames): {tot:.3f} ms, fused CvBlocks {sum(r[1] for r in rows if "cv" in r[0]):.3f} ms')
if not use_probe or os.environ.get('CP2_LIB'):
    sys.exit(0)
fn = lib().llv_dbg_cp2_stamps
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
buf = np.zeros(64 * 8 * 16, dtype=np.uint64)
LABELS3 = ['wait operands g0', 'stage A g0', 'barrier + DMA issue g0', 'stage B g0', 'wait g1', 'stage A g1', 'barrier + DMA issue g1', 'stage B g1',
           'wait g2', 'stage A g2', 'barrier + DMA issue g2', 'stage B g2', 'barrier', 'epilogue: staging + barrier', 'epilogue: stores']
LABELS1 = ['wait operands', 'stage A', 'barrier + DMA issue', 'stage B', 'barrier', 'epilogue: staging + barrier', 'epilogue: stores']
for name in ('temp1.inputcv', 'temp1.outputcv'):
    step = [s for s in plan.plan.steps if s[0] == name][0]
    assert fn(None, 1) == 0
    rc = step[1](*step[2], C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    torch.cuda.synchronize()
    assert fn(buf.ctypes.data, 1) == 0
    t = buf.reshape(64, 8, 16).astype(np.float64)
    three = name.endswith('inputcv')
    idx = list(range(13)) + [13, 14, 15] if three else [0, 1, 2, 3, 4, 13, 14, 15]
    labels = LABELS3 if three else LABELS1
    tt = t[:, 1:7, :][:, :, idx]                   # tiles 1..6 of each stamped workgroup (steady state)
    ok = (tt > 0).all(axis=2)
    d = np.diff(tt, axis=2)[ok]
    tile = (tt[:, :, -1] - tt[:, :, 0])[ok]
    print(f'{name}: {ok.sum()} stamped tiles; s_memtime ticks (100 MHz constant clock -> x10 ns) per tile: median {np.median(tile):.0f} = {np.median(tile) * 10 / 1e3:.2f} us')
    for lab, col in zip(labels, d.T):
        print(f'    {lab:32s} median {np.median(col):7.0f}  mean {col.mean():7.0f}   ({100 * col.mean() / tile.mean():4.1f} %)')
