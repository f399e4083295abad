"""Per-kernel HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs).

usage: traffic_from_pmc.py <fetch_dir> <write_dir> <frames_per_step> <out.json> [launches.json]
With launches.json (bench.py --dump-launches: the plan's steps in launch order with the device kernel each one starts) the dispatches
of every kernel are attributed to the plan's steps: the i-th dispatch of kernel K is the (i mod L)-th of the plan's L steps that
launch K (the profiled run replays ONE plan eagerly, `--shard clips` so that the calibration pass uses the same plan); a kernel whose
dispatch count is not a multiple of L is reported under "unattributed". Each step gets counter bytes next to its algorithmic bytes.
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, so it is doubled
(MI355X_MICROARCH.md, HBM section). WRITE_SIZE is taken as is."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short


SEQ = {}


def collect(d, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    seq = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            k = short(r['Kernel_Name'])
            tot[k] += float(r['Counter_Value'])
            cnt[k] += 1
            seq[k].append((int(r.get('Dispatch_Id', len(seq[k]))), float(r['Counter_Value'])))
    SEQ[counter] = {k: [v for _, v in sorted(s)] for k, s in seq.items()}
    return tot, cnt


fd, wd, fps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
ft, fc = collect(fd, 'FETCH_SIZE')
wt, wc = collect(wd, 'WRITE_SIZE')
per = {}
for k in sorted(set(ft) | set(wt)):
    n = max(fc[k], wc[k])
    if n == 0:
        continue
    fetch = 2.0 * ft[k] * 1024 / max(fc[k], 1)
    write = wt[k] * 1024 / max(wc[k], 1)
    per[k] = {'dispatches_profiled': n, 'fetch_bytes_per_launch_x2': round(fetch), 'write_bytes_per_launch': round(write),
              'hbm_bytes_per_launch': round(fetch + write)}
per_launch, unattributed = {}, []
if len(sys.argv) > 5:
    steps = json.load(open(sys.argv[5]))
    by_kernel = collections.defaultdict(list)
    for st in steps:
        if st['kernel']:
            by_kernel[st['kernel'].split('+')[0]].append(st)
    for k, sts in by_kernel.items():
        fs, ws = SEQ['FETCH_SIZE'].get(k, []), SEQ['WRITE_SIZE'].get(k, [])
        L = len(sts)
        if not fs or len(fs) % L or len(ws) != len(fs):
            unattributed.append({'kernel': k, 'steps': L, 'fetch_dispatches': len(fs), 'write_dispatches': len(ws)})
            continue
        reps = len(fs) // L
        for i, st in enumerate(sts):
            fetch = 2.0 * 1024 * sum(fs[i::L]) / reps
            write = 1024.0 * sum(ws[i::L]) / reps
            per_launch[st['name']] = {'kernel': k, 'fetch_bytes_x2': round(fetch), 'write_bytes': round(write), 'hbm_bytes': round(fetch + write),
                                      'algorithmic_bytes': round(st['bytes']), 'ratio': round((fetch + write) / st['bytes'], 3) if st['bytes'] else None}
json.dump({'frames_per_step': fps, 'per_kernel': per, 'per_launch': per_launch, 'unattributed': unattributed,
           'how': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (no trace domains) over '
                  '`python3 bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-latency-mode --no-kernel-timing`; '
                  'per kernel: sum over its dispatches / number of dispatches; FETCH_SIZE (KB) doubled per the '
                  'MI355X_MICROARCH.md HBM note for gfx950, WRITE_SIZE (KB) as is'}, open(out, 'w'), indent=1)
for k, v in sorted(per.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['dispatches_profiled'])[:12]:
    print(f"{k:44s} {v['dispatches_profiled']:5d} launches  {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch")
if per_launch:
    print('--- per launch (counter bytes / algorithmic bytes) ---')
    for n, v in per_launch.items():
        print(f"{n:34s} {v['kernel']:34s} fetch {v['fetch_bytes_x2'] / 1e6:9.1f}  write {v['write_bytes'] / 1e6:9.1f}  alg {v['algorithmic_bytes'] / 1e6:9.1f} MB  x{v['ratio']}")
    for u in unattributed:
        print('unattributed:', u)
