#!/bin/bash
# kernel trace of the single-frame plan's post-processing (decode + NMS): bash tools/post_trace.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$tag
mkdir -p $O
rocprofv3 --kernel-trace -d $O/t -o s --output-format csv -- python3 $R/tools/launch_table.py 1 > $O/log.txt 2>&1 || echo failed
python3 - <<PY
import csv, glob, re
f = glob.glob('$O/t/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# last occurrence of class_max .. pack/nms_merge
idx = [i for i, r in enumerate(rows) if 'class_max_kernel' in r['Kernel_Name']]
a = idx[-1]
t0 = int(rows[a - 2]['Start_Timestamp'])
for r in rows[a - 2:a + 16]:
    m = re.search(r'(\w+)(<[^>]*>)?\(', r['Kernel_Name'])
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  grid {r['Grid_Size_X']:>8s} x{r['Grid_Size_Y']:>3s}  wg {r['Workgroup_Size_X']:>4s}  {m.group(1) if m else r['Kernel_Name'][:40]}")
PY
