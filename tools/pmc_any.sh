#!/bin/bash
# PMC passes over any python script: bash tools/pmc_any.sh <outdir> <kernel-substring> <script> [args...]
out=$1; filt=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$out
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16" ; do
  i=$((i+1))
  rocprofv3 --pmc $pmc -d $R/gpurun_out/$out/p$i -o p --output-format csv -- python3 $R/$1 ${@:2} > $R/gpurun_out/$out/log$i.txt 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('$R/gpurun_out/$out/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
for k, d in tot.items():
    if '$filt' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:32s} {v / cnt[(k, c)]:16.1f} per launch ({cnt[(k, c)]} launches)')
PY
