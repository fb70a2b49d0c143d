#!/usr/bin/env python3
"""Developer tool: the ls-1 sweep with and without the row workspace (rows shared by several workgroups) at small batch.

    python scripts/sweep_split.py [batch ...]
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
from quant import _hip  # noqa: E402

dev = 'cuda:0'
lib = _hip.lib()
for n in [int(v) for v in sys.argv[1:]] or [1, 8, 32, 100]:
    for c, h in [(64, 56), (128, 28), (256, 14), (512, 7), (64, 32), (128, 16)]:
        x = torch.randn(n, c, h, h, device=dev)
        g = _hip.make_geom(n, c, h, h, 64, 3, 3, (1, 1), (1, 1), (1, 1), 1)
        planes = torch.zeros((_hip.act_plane_words(g),), dtype=torch.int64, device=dev)
        scales = torch.empty((1, n), device=dev)
        ws = torch.zeros((lib.lsq_sweep_workspace_bytes(n),), dtype=torch.uint8, device=dev)
        out = []
        for w in (None, ws):
            def call():
                lib.lsq_act_quant(x.data_ptr(), ctypes.byref(g), 1, 1, 3, 2.0, None, None, None, planes.data_ptr(), scales.data_ptr(),
                                  None if w is None else w.data_ptr(), 0 if w is None else w.numel(), _hip.stream_ptr(x.device))
            for _ in range(20):
                call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                call()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / 300 * 1e6)
        print(f'N={n:4d} C={c:4d} H={h:3d}: one workgroup per row {out[0]:6.1f} us, shared rows {out[1]:6.1f} us', flush=True)
