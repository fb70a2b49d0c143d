#!/usr/bin/env python3
"""Developer tool: the CIFAR-100 ResNet-18 config (batch 100, 32x32) with the XNOR convolutions on the popcount kernel
vs the dispatcher's choice (integer MFMA where eligible), per-kernel times from the instrumented step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402
from quant import _hip  # noqa: E402

dev = torch.device('cuda', 0)
for batch in (100, 256):
    model = bench.build_model(bench.cifar_arch(), dev)
    x = torch.randn(batch, 3, 32, 32, device=dev)
    for popcount in (True, False):
        _hip.xnor_impl(popcount)
        with torch.no_grad():
            for _ in range(5):
                model(x)
            torch.cuda.synchronize()
            _hip.enable_timing(True)
            model(x)
            torch.cuda.synchronize()
            t = _hip.drain_timing()
            _hip.enable_timing(False)
        print(f'batch {batch} popcount_only={popcount}: ' + '  '.join(f'{k} {v[1] * 1e3:.0f} us' for k, v in t.items()))
_hip.xnor_impl(False)
