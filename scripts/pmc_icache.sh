#!/bin/bash
# Developer tool: instruction-cache behaviour of the step's kernels (the quantizer instantiations are 250-290 KB of code against a
# 64 KB instruction cache): SQC_ICACHE_* and SQ_IFETCH* per dispatch over bench.py's headline forward on one stream, and over
# scripts/kbench.py (the same kernels launched back to back) for comparison.   usage: scripts/pmc_icache.sh <out.txt>
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=$1
dir=gpurun_out/pmc_icache
rm -rf $dir; mkdir -p $dir
p=0
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  p=$((p+1))
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $dir/net$p -- python bench.py --steps 3 --warmup 2 --min-seconds 0 --cpu-sample 0 --no-configs --no-roofline --streams 1 > $dir/net$p.log 2>&1
done
python - "$dir" > $out <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/net*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'lsq::' not in k:
            continue
        name = k[k.find('lsq::(anonymous namespace)::') + 28:].split('(')[0]
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
print('per dispatch (mean over the forwards of bench.py --streams 1): kernel | icache requests, hit rate, misses (duplicates) | ifetch, mean in flight | wave cycles, share waiting for an instruction | instructions issued')
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    g = lambda c: m.get(c, float('nan'))
    print(f"{k:44s} | req {g('SQC_ICACHE_REQ'):12.0f} hit {g('SQC_ICACHE_HITS') / max(g('SQC_ICACHE_REQ'), 1):.3f} miss {g('SQC_ICACHE_MISSES'):10.0f} (dup {g('SQC_ICACHE_MISSES_DUPLICATE'):10.0f})"
          f" | ifetch {g('SQ_IFETCH'):12.0f} level {g('SQ_IFETCH_LEVEL') / max(g('SQ_IFETCH'), 1):7.2f}"
          f" | wave cycles {g('SQ_WAVE_CYCLES'):14.0f} busy {g('SQ_BUSY_CYCLES'):12.0f} wait_inst {g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):.3f} active {g('SQ_ACTIVE_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):.3f}"
          f" | valu {g('SQ_INSTS_VALU'):12.0f} salu {g('SQ_INSTS_SALU'):12.0f}")
PY
