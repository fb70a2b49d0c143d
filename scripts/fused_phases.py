#!/usr/bin/env python3
"""Developer tool: per-phase time line of the single-launch quantizer (needs a -DLSQ_PHASE_CLOCKS build):

    make -C ml-quant_amd/csrc OUTDIR=$PWD/ml-quant_amd/lib_dbg EXTRA=-DLSQ_PHASE_CLOCKS
    LSQ_HIP_LIB=$PWD/ml-quant_amd/lib_dbg/liblsq_hip.so python scripts/fused_phases.py [--dist gauss]

Marks (100 MHz constant clock, lane 0 of every workgroup): 0 entry, 1 pass 1 + histogram done, 2-4 level-1 scan (bin
sums + block scan, prefixes of the non-empty bins, candidate test + slots), 5 keys of the flagged bins copied from the
registers to the LDS list, 6 round-0 node histograms, 7 round-0 node scans, 8 refinement done, 9 v1 known, 10 pass 2 done.
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from quant import _hip  # noqa: E402

SHAPES = [(64, 56), (128, 28), (256, 14), (512, 7)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--dist', default='gauss')
    ap.add_argument('--shapes', default='', help='e.g. 256x14,512x7 (default: the four ResNet-18 shapes)')
    ap.add_argument('--mode', type=int, default=0, help='lsq_debug_fused_mode')
    args = ap.parse_args()
    lib = _hip.lib()
    n = args.batch
    shapes = [tuple(int(v) for v in t.split('x')) for t in args.shapes.split(',')] if args.shapes else SHAPES
    lib.lsq_debug_fused_mode(args.mode)
    for c, h in shapes:
        x = torch.randn(n, c, h, h, device='cuda')
        if args.dist != 'gauss':
            x = x.clamp(min=0)
        if args.dist == 'relu-bn':
            x = x * (0.5 + torch.rand(1, c, 1, 1, device='cuda')) * 1.7 + torch.randn(1, c, 1, 1, device='cuda') * 0.5 - 0.7
        g = _hip.make_geom(n, c, h, h, c, 3, 3, (1, 1), (1, 1), (1, 1), 1)
        planes = torch.zeros(2 * _hip.act_plane_words(g), dtype=torch.int64, device='cuda')
        scales = torch.empty((2, n), device='cuda')
        for _ in range(3):
            _hip.act_quant(x, g, 2, 2, 3, 3.0, planes, scales)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16384)()
        lib.lsq_debug_read_fused_times(buf)
        t = np.array(buf, dtype=np.int64).reshape(1024, 16)[:n, :11].astype(np.float64) / 100.0   # us
        t0 = t[:, 0].min()
        raw = np.array(buf, dtype=np.int64).reshape(1024, 16)[:n]
        d = np.diff(t, axis=1)
        names = ['pass1', 'l1.sum', 'l1.pre', 'l1.test', 'copy', 'r0.hist', 'r0.scan', 'rest', 'argmin', 'pass2']
        print(f'C={c} H={h}: kernel span {t[:, 10].max() - t0:7.1f} us; entry skew {t[:, 0].max() - t0:5.1f} us; per-phase median / max (us): ' +
              '  '.join(f'{nm} {np.median(d[:, i]):5.1f}/{d[:, i].max():5.1f}' for i, nm in enumerate(names)))
        r1 = (raw[:, 11:13].astype(np.float64) / 100.0)
        print(f'      round 1: list pass {np.median(r1[:, 0] - t[:, 7]):5.1f} us, tasks/nodes {np.median(r1[:, 1] - r1[:, 0]):5.1f} us, after {np.median(t[:, 8] - r1[:, 1]):5.1f} us;'
              f' listed keys median {int(np.median(raw[:, 13]))} max {raw[:, 13].max()}, tasks median {int(np.median(raw[:, 14]))} max {raw[:, 14].max()},'
              f' flagged bins median {int(np.median(raw[:, 15] % 1000))}, nodes median {int(np.median(raw[:, 15] // 1000))} max {(raw[:, 15] // 1000).max()}')


if __name__ == '__main__':
    main()
