#!/usr/bin/env python3
"""Developer tool: scripts/fused_phases.py for BOTH row layouts side by side (needs a -DLSQ_PHASE_CLOCKS build):
per-phase time line of the single-launch quantizer on NCHW rows and on three-stream rows of the same values."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import numpy as np
import torch
from quant import _hip
from quant.binary import layouts

lib = _hip.lib()
n = 256
names = ['pass1', 'S1', 'S2', 'S3', 'sweep', 'rank', 'x', 'x', 'argmin', 'pass2']
for c, h in [(64, 56), (128, 28), (256, 14), (512, 7)]:
    x = torch.randn(n, c, h, h, device='cuda')
    g = _hip.make_geom(n, c, h, h, c, 3, 3, (1, 1), (1, 1), (1, 1), 1)
    planes = torch.zeros(2 * _hip.act_plane_words(g), dtype=torch.int64, device='cuda')
    scales = torch.empty((2, n), device='cuda')
    xs = layouts.info(layouts.from_nchw(x)).buf
    for tag, src, lay, fn in (('nchw', x, 0, lib.lsq_debug_read_fused_times), ('s3  ', xs, 1, lib.lsq_debug_read_fused_times_s3)):
        for _ in range(3):
            _hip.act_quant(src, g, 2, 2, 3, 3.0, planes, scales, None, None, lay)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            _hip.act_quant(src, g, 2, 2, 3, 3.0, planes, scales, None, None, lay)
        e1.record()
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16384)()
        fn(buf)
        t = np.array(buf, dtype=np.int64).reshape(1024, 16)[:n, :11].astype(np.float64) / 100.0
        d = np.diff(t, axis=1)
        print(f'C={c} H={h} {tag}: {1e3 * e0.elapsed_time(e1) / 10:6.1f} us/launch; span {t[:, 10].max() - t[:, 0].min():6.1f}; ' +
              '  '.join(f'{nm} {np.median(d[:, i]):5.1f}' for i, nm in enumerate(names) if nm != 'x'))
