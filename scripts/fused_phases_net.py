#!/usr/bin/env python3
"""Developer tool: phase time line of the single-launch quantizer for every QuantConv2d of the real ResNet-18
forward (needs a -DLSQ_PHASE_CLOCKS build, see scripts/fused_phases.py):

    LSQ_HIP_LIB=$PWD/ml-quant_amd/lib_dbg/liblsq_hip.so python scripts/fused_phases_net.py [--batch 256]
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
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    model = bench.build_model(bench.imagenet_arch(), dev)
    x = torch.randn(args.batch, 3, 224, 224, generator=torch.Generator().manual_seed(0)).to(dev)
    lib = _hip.lib()
    orig = _hip.act_quant
    rows = []

    def wrapped(xx, geom, *a, **k):
        orig(xx, geom, *a, **k)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16384)()
        lib.lsq_debug_read_fused_times(buf)
        raw = np.array(buf, dtype=np.int64).reshape(1024, 16)[:geom.N]
        rows.append(((geom.C, geom.H), raw.copy()))

    with torch.no_grad():
        model(x)
        _hip.act_quant = wrapped
        model(x)
        _hip.act_quant = orig
    names = ['pass1', 'l1', 'copy', 'r0', 'rest', 'slow+argmin', 'pass2']
    cuts = [0, 1, 4, 5, 7, 8, 9, 10]
    tot = 0.0
    for i, ((c, h), raw) in enumerate(rows):
        t = raw[:, :11].astype(np.float64) / 100.0
        span = t[:, 10].max() - t[:, 0].min()
        tot += span
        seg = [np.median(t[:, cuts[k + 1]] - t[:, cuts[k]]) for k in range(len(names))]
        segmax = [np.max(t[:, cuts[k + 1]] - t[:, cuts[k]]) for k in range(len(names))]
        r1 = raw[:, 11:13].astype(np.float64) / 100.0
        extra = f' | round1: list {np.median(r1[:, 0] - t[:, 7]):4.1f} work {np.median(r1[:, 1] - r1[:, 0]):4.1f} after {np.median(t[:, 8] - r1[:, 1]):4.1f}/{np.max(t[:, 8] - r1[:, 1]):4.1f}'
        print(f'layer {i:2d} C={c:3d} H={h:2d}: span {span:6.1f} us | ' + '  '.join(f'{nm} {a:5.1f}/{b:5.1f}' for nm, a, b in zip(names, seg, segmax)) +
              f' | listed {int(np.median(raw[:, 13]))}/{raw[:, 13].max()} tasks {int(np.median(raw[:, 14]))}/{raw[:, 14].max()}'
              f' bins {int(np.median(raw[:, 15] % 1000))}/{(raw[:, 15] % 1000).max()} nodes {int(np.median(raw[:, 15] // 1000))}/{(raw[:, 15] // 1000).max()}' + extra)
    print(f'sum of kernel spans: {tot / 1e3:.3f} ms')


if __name__ == '__main__':
    main()
