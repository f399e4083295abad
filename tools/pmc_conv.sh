#!/bin/bash
# PMC passes over one conv shape (tools/ab_conv.py <tile> <shape>): bash tools/pmc_conv.sh <tile> <shape> <outdir>
# Counters only (no trace domains), program directly after "--".
tile=$1; shape=$2; out=$3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $pmc -d $R/gpurun_out/$out/p$i -o p --output-format csv -- python3 $R/tools/ab_conv.py $tile $shape > $R/gpurun_out/$out/log$i.txt 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('$R/gpurun_out/$out/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); 
        cnt[(k, r['Counter_Name'])] += 1
for k, d in tot.items():
    if 'conv' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:32s} {v / cnt[(k, c)]:16.1f} per launch ({cnt[(k, c)]} launches)')
PY
