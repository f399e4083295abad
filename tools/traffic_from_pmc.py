"""Per-kernel HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs).

usage: traffic_from_pmc.py <fetch_dir> <write_dir> <frames_per_step> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, so it is doubled
(MI355X_MICROARCH.md, HBM section). WRITE_SIZE is taken as is."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short


def collect(d, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            k = short(r['Kernel_Name'])
            tot[k] += float(r['Counter_Value'])
            cnt[k] += 1
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
json.dump({'frames_per_step': fps, 'per_kernel': per,
           'how': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (no trace domains) over '
                  '`python3 bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-latency-mode --no-kernel-timing`; '
                  'per kernel: sum over its dispatches / number of dispatches; FETCH_SIZE (KB) doubled per the '
                  'MI355X_MICROARCH.md HBM note for gfx950, WRITE_SIZE (KB) as is'}, open(out, 'w'), indent=1)
for k, v in sorted(per.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['dispatches_profiled'])[:12]:
    print(f"{k:44s} {v['dispatches_profiled']:5d} launches  {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch")
