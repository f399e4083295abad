#!/bin/bash
# Kernel-time summary + HBM traffic of the benchmark step. Run on the GPU box:  bash tools/collect_profile.sh r01c
# 1) rocprofv3 --kernel-trace --stats over the default bench (graph replay)  -> profiles/<tag>_kernel_stats.csv
# 2) two PMC passes (FETCH_SIZE, WRITE_SIZE; counters only) over an eager run -> profiles/traffic.json
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$tag
mkdir -p $O $R/profiles
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency-mode > $O/bench_stats.log 2>&1 || echo "stats pass failed"
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 $R/bench.py --no-graph --shard clips --dump-launches $O/launches.json --steps 2 --warmup 1 --no-cpu-baseline --no-latency-mode --no-kernel-timing > $O/bench_fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 $R/bench.py --no-graph --shard clips --steps 2 --warmup 1 --no-cpu-baseline --no-latency-mode --no-kernel-timing > $O/bench_write.log 2>&1 || echo "write pass failed"
f=$(find $O/stats -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv
python3 $R/tools/traffic_from_pmc.py $O/fetch $O/write 16 $R/gpurun_out/${tag}_traffic.json $O/launches.json > $R/gpurun_out/${tag}_traffic.txt
cat $R/gpurun_out/${tag}_traffic.txt
tail -1 $O/bench_stats.log
