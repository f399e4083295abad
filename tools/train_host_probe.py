"""Where the host is while the training step runs: host time of every HIP-graph replay of bench.py --mode train (the step is a
forward graph, the loss launches, then one data-gradient and one weight-gradient graph per all-reduce bucket on two streams).
python tools/train_host_probe.py [bench args]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

LOG = []
_replay = torch.cuda.CUDAGraph.replay


def replay(self):
    t0 = time.perf_counter()
    _replay(self)
    LOG.append((t0, time.perf_counter() - t0))


torch.cuda.CUDAGraph.replay = replay
import bench  # noqa: E402

sys.argv = ['bench.py', '--mode', 'train', '--steps', '20', '--warmup', '5'] + sys.argv[1:]
bench.main()
per = [d for _, d in LOG]
n = len(per)
print('replays', n, 'host us per replay: mean %.1f max %.1f' % (1e6 * sum(per) / n, 1e6 * max(per)))
k = n // 25
last = LOG[-k:]
t0 = last[0][0]
print('last step, %d replays: start offset us / duration us' % k)
print(' '.join('%.0f/%.0f' % (1e6 * (t - t0), 1e6 * d) for t, d in last))
