#!/bin/bash
# rocprofv3 kernel-trace summary of one bench mode:  bash tools/prof_mode.sh <tag> <bench args...>
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$tag
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py "$@" > $O/bench.log 2>&1 || echo "stats pass failed"
f=$(find $O/stats -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv
grep '"metric"' $O/bench.log | cut -c1-600
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$R/gpurun_out/${tag}_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6, 'kernels', len(rows), 'calls', sum(int(r['Calls']) for r in rows))
for r in rows[:40]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
PY
