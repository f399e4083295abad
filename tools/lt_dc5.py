"""Per-launch times of the DC5 ResNet-50 (the SELSA configs' backbone) on one 736x1280 frame. Usage: python tools/lt_dc5.py"""
import sys, torch
sys.path.insert(0, '.')
from lowlightenvironmentvideoobjectdetection_amd import engine
import lowlightenvironmentvideoobjectdetection_amd.models as M
from lowlightenvironmentvideoobjectdetection_amd import registry
from lowlightenvironmentvideoobjectdetection_amd.synthetic import seeded_weights
dev = torch.device('cuda:0')
m = seeded_weights(registry.build_backbone(dict(type='ResNet', depth=50, num_stages=4, out_indices=(3,), strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2),
                                                frozen_stages=-1, norm_eval=True, style='pytorch')), seed=0)
b = engine.Builder(dev)
x_in = b.buf(1, 736, 1280, 8)
engine.build_resnet(b, engine._cpu_sd(m.state_dict()), x_in, fuse_blocks=True, **m.plan_kwargs())
b.plan.run(); torch.cuda.synchronize()
t = b.plan.run_timed(iters=10)
print('total', sum(r[1] for r in t))
for name, ms, fl, nb in t:
    print(f'{name:40s} {ms * 1e3:8.1f} us {fl / ms / 1e9 if ms > 0 else 0:8.1f} TF/s  {b.plan.kernel_of.get(name, "")}')
