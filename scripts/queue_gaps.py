#!/usr/bin/env python3
"""Developer tool: idle stretches of every HIP queue in a rocprofv3 --kernel-trace CSV (gaps between consecutive kernels of one
queue longer than `min_us`), with the kernels either side and what the other queues ran at the moment the gap ended.
    python scripts/queue_gaps.py <kernel_trace.csv> [min_us] [skip_first_ms]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
skip_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t00 = int(rows[0]['Start_Timestamp'])


def short(k):
    return (k[k.find('lsq::(anonymous namespace)::') + 28:] if 'lsq::(anon' in k else k)[:36]


byq = collections.defaultdict(list)
for r in rows:
    byq[r['Queue_Id']].append(r)
print('kernels per queue:', {q: len(v) for q, v in byq.items()}, ' span %.1f ms' % ((int(rows[-1]['End_Timestamp']) - t00) / 1e6))
for q, v in sorted(byq.items()):
    gaps = []
    for a, b in zip(v, v[1:]):
        g = (int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3
        if g >= min_us and (int(b['Start_Timestamp']) - t00) / 1e6 >= skip_ms:
            t = int(b['Start_Timestamp'])
            others = [short(o['Kernel_Name']) + ' (q%s, started %.0f us earlier)' % (o['Queue_Id'], (t - int(o['Start_Timestamp'])) / 1e3)
                      for o in rows if o['Queue_Id'] != q and int(o['Start_Timestamp']) <= t < int(o['End_Timestamp'])]
            gaps.append((g, (t - t00) / 1e3, short(a['Kernel_Name']), short(b['Kernel_Name']), others))
    tot = sum(g[0] for g in gaps)
    print(f'queue {q}: {len(gaps)} gaps >= {min_us:.0f} us, {tot / 1e3:.2f} ms in all')
    for g, t, ka, kb, others in gaps[:14]:
        print(f'   {g:8.1f} us before t = {t:9.1f} us: {ka} -> {kb} | running then: {others}')

# mean idle time of a queue between the classifier of one forward and the first kernel of the next one
import statistics
for q, v in sorted(byq.items()):
    g = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(v, v[1:])
         if a['Kernel_Name'].startswith('Cijk') and (int(b['Start_Timestamp']) - t00) / 1e6 >= skip_ms]
    if len(g) > 3:
        print(f'queue {q}: classifier -> next kernel ({short(v[-1]["Kernel_Name"])} ...): median idle {statistics.median(g):.0f} us over {len(g)} forwards')

# ... and in front of the stem, whatever precedes it
for q, v in sorted(byq.items()):
    g = [((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3, short(a['Kernel_Name'])) for a, b in zip(v, v[1:])
         if 'stem_conv' in b['Kernel_Name'] and (int(b['Start_Timestamp']) - t00) / 1e6 >= skip_ms]
    if len(g) > 3:
        print(f'queue {q}: {g[0][1]} -> stem: median idle {statistics.median(x[0] for x in g):.0f} us over {len(g)} forwards')
