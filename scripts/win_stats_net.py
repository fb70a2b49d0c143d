#!/usr/bin/env python3
"""Developer tool (round 5): the windowed level-1 solve on the rows of the real ResNet-18 forward -- how many rows it
solves, why the others fall back, phase times -- and its scales / planes against the round-2 solve (lsq_debug_fused_mode 4)
on the same inputs, bit for bit.  Needs a -DLSQ_PHASE_CLOCKS build:

    make -C ml-quant_amd/csrc OUTDIR=$PWD/ml-quant_amd/lib_dbg EXTRA=-DLSQ_PHASE_CLOCKS
    LSQ_HIP_LIB=$PWD/ml-quant_amd/lib_dbg/liblsq_hip.so python scripts/win_stats_net.py [--batch 256] [--act ls-2]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from quant import _hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--act', default='ls-2')
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    model = bench.build_model(bench.imagenet_arch(args.act, 3 if args.act == 'ls-2' else 2), dev)
    x = torch.randn(args.batch, 3, 224, 224, generator=torch.Generator().manual_seed(args.seed)).to(dev)
    lib = _hip.lib()
    orig = _hip.act_quant
    rows = []

    def wrapped(xx, geom, scheme, k, skip, alpha, planes, scales, forced=None, pre=None):
        lib.lsq_debug_fused_mode(4)
        p4, s4 = torch.zeros_like(planes), torch.zeros_like(scales)
        orig(xx, geom, scheme, k, skip, alpha, p4, s4, forced, pre)
        torch.cuda.synchronize()
        buf4 = (ctypes.c_longlong * 16384)()
        lib.lsq_debug_read_fused_times(buf4)
        raw4 = np.array(buf4, dtype=np.int64).reshape(1024, 16)[:geom.N].copy()
        lib.lsq_debug_fused_mode(0)
        orig(xx, geom, scheme, k, skip, alpha, planes, scales, forced, pre)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16384)()
        lib.lsq_debug_read_fused_times(buf)
        raw = np.array(buf, dtype=np.int64).reshape(1024, 16)[:geom.N].copy()
        wb = (ctypes.c_int * 4096)()
        lib.lsq_debug_read_win_stats(wb)
        ws = np.array(wb, dtype=np.int64).reshape(1024, 4)[:geom.N].copy()
        same = bool(torch.equal(planes, p4)) and bool(torch.equal(scales, s4))
        rows.append(((geom.C, geom.H), raw, ws, same, int((scales[0] != s4[0]).sum()), raw4))

    with torch.no_grad():
        model(x)
        _hip.act_quant = wrapped
        model(x)
        _hip.act_quant = orig
    tot = 0.0
    tot4 = 0.0
    for i, ((c, h), raw, ws, same, nv1, raw4) in enumerate(rows):
        t4 = raw4[:, :11].astype(np.float64) / 100.0
        span4 = t4[:, 10].max() - t4[:, 0].min()
        tot4 += span4
        t = raw[:, :11].astype(np.float64) / 100.0
        span = t[:, 10].max() - t[:, 0].min()
        tot += span
        ok = ws[:, 0] == 1
        names = ['pass1', 'S1', 'S2', 'S3', 'S4', 'S5', 'argmin', 'pass2']
        cuts = [0, 1, 2, 3, 4, 5, 8, 9, 10]
        seg = [np.median((t[:, cuts[k + 1]] - t[:, cuts[k]])[ok]) if ok.any() else float('nan') for k in range(len(names))]
        why = {}
        for f in ws[~ok, 1]:
            why[int(f)] = why.get(int(f), 0) + 1
        print(f'layer {i:2d} C={c:3d} H={h:2d}: span {span:6.1f} us (round-2 solve: {span4:6.1f}, its pass 1 / pass 2 {np.median(t4[:, 1] - t4[:, 0]):4.1f} / {np.median(t4[:, 10] - t4[:, 9]):4.1f}) | windowed rows {int(ok.sum())}/{len(ok)} (fall-back flags {why}) | '
              f'groups {int(np.median(ws[:, 2]))}/{ws[:, 2].max()} fine bins {int(np.median(ws[:, 3]))}/{ws[:, 3].max()} | '
              + '  '.join(f'{nm} {a:4.1f}' for nm, a in zip(names, seg)) + f' | equals the round-2 solve: {same} (v1 differs in {nv1} rows)')
    print(f'sum of kernel spans: {tot / 1e3:.3f} ms (round-2 solve: {tot4 / 1e3:.3f} ms)')


if __name__ == '__main__':
    main()
