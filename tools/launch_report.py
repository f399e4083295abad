"""Per-launch table from bench.py --dump-launches (with kernel timing): name, kernel, us, TFLOP/s, algorithmic GB/s.
usage: launch_report.py launches.json [traffic.json]   (traffic.json: tools/traffic_from_pmc.py output with per_launch)"""
import json, sys
L = json.load(open(sys.argv[1]))
T = json.load(open(sys.argv[2]))['per_launch'] if len(sys.argv) > 2 else {}
tot = sum(x.get('ms', 0) for x in L)
print(f'{"launch":34s} {"kernel":34s} {"us":>8s} {"TF/s":>7s} {"alg GB/s":>9s} {"pmc GB/s":>9s}  share')
for x in L:
    ms = x.get('ms', 0)
    if ms <= 0:
        continue
    t = T.get(x['name'])
    pmc = f"{t['hbm_bytes'] / ms / 1e6:9.0f}" if t else ' ' * 9
    print(f"{x['name']:34s} {x['kernel']:34s} {ms * 1e3:8.1f} {x['flops'] / ms / 1e9:7.0f} {x['bytes'] / ms / 1e6:9.0f} {pmc}  {ms / tot:5.1%}")
print(f'total {tot:.3f} ms')
