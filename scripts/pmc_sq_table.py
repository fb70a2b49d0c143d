#!/usr/bin/env python3
"""Developer tool: turn the SQ counter passes of scripts/pmc_kernel.sh into the JSON table bench.py reads
(profiles/rNN_pmc_sq.json): per kernel and layer shape the matrix-core utilisation
SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), VALU and LDS instructions per MFMA, LDS bank-conflict share
and the wait fractions of the wave cycles.

    python scripts/pmc_sq_table.py OUT.json  ENTRY:SHAPE:COUNT:PATTERN:DIR ...
      ENTRY   C-ABI entry point (lsq_xnor_conv2d ...), SHAPE a label (C64_H56_s1), COUNT launches of that shape per forward,
      PATTERN substring of the kernel name, DIR the gpurun_out/pmc_<tag> directory pmc_kernel.sh wrote
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ml-quant_amd'))
from quant import _hip  # noqa: E402


def counters(directory, pattern):
    agg = collections.defaultdict(list)
    for f in glob.glob(directory + '/p*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if pattern in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}, (max(len(v) for v in agg.values()) if agg else 0)


def main():
    out_path, rows = sys.argv[1], []
    for spec in sys.argv[2:]:
        entry, shape, count, pattern, directory = spec.split(':')
        c, n = counters(directory, pattern)
        if not c:
            print('no counters for', spec, file=sys.stderr)
            continue
        mfma = max(c.get('SQ_INSTS_MFMA', 0.0), 1.0)
        wave = max(c.get('SQ_WAVE_CYCLES', 0.0), 1.0)
        rows.append({
            'entry': entry, 'shape': shape, 'count_in_forward': int(count), 'kernel_pattern': pattern, 'dispatches_averaged': n,
            'busy_cu_cycles': c.get('SQ_BUSY_CU_CYCLES', 0.0),
            'mfma_busy_frac': c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / max(4.0 * c.get('SQ_BUSY_CU_CYCLES', 0.0), 1.0),
            'insts_mfma': c.get('SQ_INSTS_MFMA', 0.0), 'valu_per_mfma': (c.get('SQ_INSTS_VALU', 0.0) - c.get('SQ_INSTS_MFMA', 0.0)) / mfma,
            'salu_per_mfma': c.get('SQ_INSTS_SALU', 0.0) / mfma, 'lds_per_mfma': c.get('SQ_INSTS_LDS', 0.0) / mfma,
            'vmem_rd_per_mfma': c.get('SQ_INSTS_VMEM_RD', 0.0) / mfma, 'vmem_wr_per_mfma': c.get('SQ_INSTS_VMEM_WR', 0.0) / mfma,
            'lds_bank_conflict_frac': c.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(c.get('SQ_LDS_IDX_ACTIVE', 0.0), 1.0),
            'wait_any_frac': c.get('SQ_WAIT_ANY', 0.0) / wave, 'wait_inst_any_frac': c.get('SQ_WAIT_INST_ANY', 0.0) / wave,
            'active_inst_any_frac': c.get('SQ_ACTIVE_INST_ANY', 0.0) / wave, 'waves': c.get('SQ_WAVES', 0.0),
        })
    json.dump({'note': 'rocprofv3 --pmc passes (scripts/pmc_kernel.sh), one kernel on its own at batch 256; mfma_busy_frac = '
                       'SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)', 'csrc_sha256': _hip.source_fingerprint(), 'kernels': rows}, open(out_path, 'w'), indent=1)
    for r in rows:
        print(f"{r['entry']:18s} {r['shape']:12s} mfma busy {r['mfma_busy_frac']:.3f}  VALU/MFMA {r['valu_per_mfma']:.1f}  LDS conflicts "
              f"{r['lds_bank_conflict_frac']:.2f}  wait {r['wait_any_frac']:.2f}/{r['wait_inst_any_frac']:.2f}")


if __name__ == '__main__':
    main()
