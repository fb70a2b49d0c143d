#!/bin/bash
# HBM traffic per kernel from the L2 fabric-side counters, collected as MI355X_MICROARCH.md (HBM section)
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel-trace only), units KB,
# FETCH_SIZE doubled on gfx950.  Workload: bench.py itself (ResNet-18 headline config, batch 256, the real
# activations and folded batch norms), a few steps.
# usage: scripts/pmc_traffic.sh <out.json> [bench args]
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=$1; shift
dir=gpurun_out/pmc_traffic
rm -rf $dir; mkdir -p $dir
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $dir/$c -- python bench.py --steps 3 --warmup 2 --min-seconds 0 --cpu-sample 0 --no-configs --no-roofline --streams 1 "$@" > $dir/$c.log 2>&1
done
python - "$dir" "$out" <<'PY'
import collections, csv, glob, json, sys
sys.path.insert(0, 'ml-quant_amd')
from quant import _hip
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'lsq::' not in k:
            continue
        name = k[k.find('lsq::(anonymous namespace)::') + 28:].split('(')[0]
        agg[f"{name} grid={r['Grid_Size']}"][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, d in sorted(agg.items()):
    f = sum(d['FETCH_SIZE']) / max(len(d['FETCH_SIZE']), 1)
    w = sum(d['WRITE_SIZE']) / max(len(d['WRITE_SIZE']), 1)
    res[k] = {'FETCH_SIZE_KB': round(f, 1), 'WRITE_SIZE_KB': round(w, 1),
              'hbm_read_MB_corrected': round(2 * f / 1024, 1), 'hbm_write_MB': round(w / 1024, 1)}
json.dump({'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over '
                   'bench.py (ResNet-18 ls-1w/ls-2a, batch 256, 5 forwards); per-dispatch averages over all layers a kernel serves; grid = threads; FETCH_SIZE doubled per '
                   'MI355X_MICROARCH.md (gfx950 counts 128-byte requests as 64 bytes)', 'csrc_sha256': _hip.source_fingerprint(), 'kernels': res},
          open(sys.argv[2], 'w'), indent=1)
print(open(sys.argv[2]).read()[:3000])
PY
