#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per-kernel time per step over the LAST `steps`
forward passes (MIOpen's first-run search kernels in the warm-up are excluded).

    python scripts/trace_summary.py <bench_kernel_trace.csv> [steps]
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'MeanOps' in r['Kernel_Name']]          # avg-pool: once per forward
start, end = ends[-steps - 1] + 1, ends[-1] + 1
agg = collections.defaultdict(lambda: [0, 0])
for r in rows[start:end]:
    k = r['Kernel_Name']
    name = k[k.find('lsq::(anonymous namespace)::') + 28:][:40] if 'lsq::' in k else k[:70]
    agg[name][0] += 1
    agg[name][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(v[1] for v in agg.values())
span = int(rows[end - 1]['End_Timestamp']) - int(rows[start]['Start_Timestamp'])
print(f'steps analysed: {steps}; kernel time per step {tot / steps / 1e6:.3f} ms; wall span per step {span / steps / 1e6:.3f} ms')
print('us_per_step,launches_per_step,kernel')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{v[1] / steps / 1e3:.1f},{v[0] / steps:.1f},"{k}"')
