#!/bin/bash
# Developer tool: SQ counters of the kernels whose name contains <pattern> while running <command>
# (separate passes, kernel-trace only).  usage: scripts/pmc_kernel.sh <pattern> <outdir-tag> <command...>
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
pat=$1; tag=$2; shift 2
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1
done
python - "$out" "$pat" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']:
            agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})')
PY
