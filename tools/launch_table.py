import sys, torch, json
sys.path.insert(0, '/root/repo')
import bench
from lowlightenvironmentvideoobjectdetection_amd import data, engine
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
det, cleaner = bench.build_models(0)
clean = data.synth_clean_clip(max(8, B), 720, 1280, seed=100)
clip, img_shape, pad_shape = data.preprocess_clip(data.darken_and_add_noise(clean, seed=200), dev)
from lowlightenvironmentvideoobjectdetection_amd.synthetic import calibrate_head_on_device
plan = engine.EnhanceDetectPlan(cleaner.state_dict(), det.state_dict(), 80, B, 736, 1280, dev, use_graph=False)
plan.den.frames.copy_(clip[:B + 4])
plan.plan.run(); torch.cuda.synchronize()
calibrate_head_on_device(det, plan.det, target_cls_mean=-6.0, target_cls_std=1.3)
plan = engine.EnhanceDetectPlan(cleaner.state_dict(), det.state_dict(), 80, B, 736, 1280, dev, use_graph=False)
plan.den.frames.copy_(clip[:B + 4])
plan.plan.run(); torch.cuda.synchronize()
print('candidates above thr:', int((plan.det.post.dp.scores.reshape(B, -1)[0] > 0.05).sum()), 'dets:', int(plan.det.post.nb.count[0].item()))
t = plan.plan.run_timed(iters=5)
tot = sum(x[1] for x in t)
print(f'total {tot:.3f} ms, flops {plan.plan.flops/1e9:.1f} G')
agg = {}
for name, ms, fl, nb in t:
    print(f'{name:45s} {ms*1e3:9.1f} us  {fl/ms/1e9 if ms>0 else 0:8.1f} TF/s  {nb/ms/1e6 if ms>0 else 0:8.1f} GB/s  {fl/1e9:8.2f} GF')
    key = name.split('.')[0] if not name.startswith('backbone') else '.'.join(name.split('.')[:2])
    a = agg.setdefault(key, [0, 0]); a[0] += ms; a[1] += fl
for k, (ms, fl) in agg.items():
    print(f'== {k:30s} {ms:8.3f} ms {fl/ms/1e9 if ms > 0 else 0:8.1f} TF/s {fl/1e9:9.1f} GF')
