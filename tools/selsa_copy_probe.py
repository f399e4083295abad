"""Where do the device-to-device copies of `bench.py --mode selsa` come from? Runs the bench's model for a few key frames with torch's
copying entry points wrapped (copy_, clone, contiguous-that-copies, cat, stack, to, index_select ...) and prints the Python call sites
that issued them in ONE steady-state step, most frequent first.  python tools/selsa_copy_probe.py  (on the GPU box)"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import bench
    import argparse
    sites = collections.Counter()
    active = [False]

    def wrap(owner, name):
        orig = getattr(owner, name)

        def f(*a, **k):
            if active[0]:
                fr = [l for l in traceback.extract_stack()[:-1] if 'lowlightenvironment' in l.filename or l.filename.endswith('bench.py')]
                if fr:
                    sites[(name, os.path.basename(fr[-1].filename), fr[-1].lineno)] += 1
            return orig(*a, **k)
        setattr(owner, name, f)
    for n in ('copy_', 'clone', 'contiguous', 'to', 'float', 'half', 'index_select', 'repeat', 'expand_as', '__getitem__', '__setitem__'):
        wrap(torch.Tensor, n)
    for n in ('cat', 'stack', 'zeros', 'empty', 'zeros_like', 'empty_like', 'full', 'tensor', 'arange'):
        wrap(torch, n)
    args = argparse.Namespace(height=720, width=1280, steps=6, warmup=2, gpus=1)
    dev = torch.device('cuda:0')
    # the bench's own step function, instrumented through its closure: run warm-up outside, one step inside the counter
    real_sync = torch.cuda.synchronize
    calls = [0]

    def sync_hook(*a, **k):
        calls[0] += 1
        if calls[0] == 1:          # the synchronize between warm-up and the timed loop
            active[0] = True
        return real_sync(*a, **k)
    torch.cuda.synchronize = sync_hook
    bench.selsa_bench(args, 1, 0, dev)
    active[0] = False
    print(f'--- copying entry points called from the package during the {args.steps} timed steps (per step = / {args.steps}) ---')
    for (name, fn, ln), c in sites.most_common(40):
        print(f'{c / args.steps:7.1f} per step  {name:14s} {fn}:{ln}')


if __name__ == '__main__':
    main()
