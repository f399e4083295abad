#!/bin/bash
# same-box A/B of two builds of the library: libllv_ab.so (A) against libllv.so (B), default bench, twice each
P=lowlightenvironmentvideoobjectdetection_amd
cp $P/libllv.so /tmp/new.so; cp $P/libllv_ab.so /tmp/old.so
for r in 1 2; do
for v in old new; do
  cp /tmp/$v.so $P/libllv.so
  timeout -k 10 400 python bench.py --no-cpu-baseline 2>&1 | grep metric > gpurun_out/ab_$v.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v.json").read())
pk=d["roofline"]["per_kernel"]
print("$v", d["value"], "lat", d["latency_mode"]["value"], "p3", d["streaming_pipelined3_mode"]["value"], " ".join(f"{k.split('_kernel')[0][-14:]}{k.split('_kernel')[1]}={v['ms_per_step']}" for k,v in pk.items()))
PY
done
done
cp /tmp/new.so $P/libllv.so
