#!/usr/bin/env python3
"""Developer tool: lsq_stem_conv_pool at batch 256, 224 x 224, for the three operand splits (3: bf16 x 3, six MFMA
passes; 2: bf16 x 2; 22: fp16 + scaled fp16 remainder, three passes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
from quant import _hip  # noqa: E402

x = torch.randn(256, 3, 224, 224, device='cuda')
w = torch.randn(64, 3, 7, 7, device='cuda') * 0.05
b = torch.randn(64, device='cuda')
for split in (3, 2, 22):
    for _ in range(3):
        _hip.stem_conv_pool(x, w, b, split)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _hip.stem_conv_pool(x, w, b, split)
    e1.record()
    torch.cuda.synchronize()
    print(f'split {split}: {e0.elapsed_time(e1) * 100:.1f} us')
