#!/usr/bin/env python3
"""Developer tool (round 5): what runs UNDER what when consecutive batches alternate between HIP streams.  Input: a rocprofv3
--kernel-trace CSV of `bench.py --streams 2` (two-stream region first, one-stream region after it); `steps` forwards of each: how long the chip
had 0 / 1 / 2+ kernels in flight, per-kernel mean duration there against the same kernel in the single-stream region, and a
timeline of one step.
    python scripts/overlap_trace.py <kernel_trace.csv> [steps] [timeline_rows]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nline = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(k):
    return k[k.find('lsq::(anonymous namespace)::') + 28:][:44] if 'lsq::' in k else k[:44]


ends = [i for i, r in enumerate(rows) if 'MeanOps' in r['Kernel_Name']]          # avg-pool: once per forward


def region(lo, hi):
    sel = rows[lo:hi]
    t0 = min(int(r['Start_Timestamp']) for r in sel)
    t1 = max(int(r['End_Timestamp']) for r in sel)
    ev = []
    for r in sel:
        ev.append((int(r['Start_Timestamp']), 1))
        ev.append((int(r['End_Timestamp']), -1))
    ev.sort()
    depth, last, hist = 0, t0, collections.Counter()
    for t, d in ev:
        hist[min(depth, 3)] += t - last
        last = t
        depth += d
    per = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        per[short(r['Kernel_Name'])][0] += 1
        per[short(r['Kernel_Name'])][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    return t1 - t0, hist, per


# bench.py times the two-stream region FIRST (the headline) and the one-stream region after it: take `steps` forwards from
# the middle of each, and tell them apart by the queues they ran on
n_fw = len(ends)
first = (ends[n_fw // 4] + 1, ends[n_fw // 4 + steps] + 1)
last = (ends[-steps - 1] + 1, ends[-1] + 1)
nq = lambda r: len({row.get('Queue_Id', '?') for row in rows[r[0]:r[1]]})      # noqa: E731
piped, single = (first, last) if nq(first) >= nq(last) else (last, first)
piped_lo, piped_hi = piped
span_p, hist_p, per_p = region(piped_lo, piped_hi)
span_s, hist_s, per_s = region(*single)
for tag, span, hist in (('single-stream region', span_s, hist_s), ('pipelined region', span_p, hist_p)):
    tot = sum(hist.values())
    print(f'{tag}: {span / steps / 1e6:.3f} ms per step; kernels in flight 0 / 1 / 2 / 3+: ' +
          ' / '.join(f'{100 * hist[d] / tot:.1f} %' for d in range(4)))
print('kernel, launches per step, us per launch single-stream, us per launch pipelined')
for k in sorted(per_s, key=lambda k: -per_s[k][1]):
    a, b = per_s[k], per_p.get(k, [0, 0])
    print(f'{k:46s} {a[0] / steps:5.1f} {a[1] / max(a[0], 1) / 1e3:8.1f} {b[1] / max(b[0], 1) / 1e3:8.1f}')
print('sum of kernel durations per step: single %.3f ms, pipelined %.3f ms' % (
    sum(v[1] for v in per_s.values()) / steps / 1e6, sum(v[1] for v in per_p.values()) / steps / 1e6))
print('queues of the pipelined region:', dict(collections.Counter(r.get('Queue_Id', '?') for r in rows[piped_lo:piped_hi])))
print('timeline (pipelined region, us from the first row): start, duration, queue, kernel')
base = int(rows[piped_lo]['Start_Timestamp'])
for r in rows[piped_lo:piped_lo + nline]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f'{(s - base) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{r.get("Queue_Id", "?")}  {short(r["Kernel_Name"])}')
