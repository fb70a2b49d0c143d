#!/usr/bin/env python3
"""Developer tool: the fused stem kernel against the MIOpen convolution + tail kernel path it replaces (time and
difference), batch 256 at 224x224.

    python scripts/stem_one.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd'), os.path.join(ROOT, 'scripts')]
import torch  # noqa: E402
import bench  # noqa: E402
import quant.models.resnet as R  # noqa: E402
from kbench import timeit  # noqa: E402

dev = torch.device('cuda', 0)
model = bench.build_model(bench.imagenet_arch(), dev)
x = torch.randn(256, 3, 224, 224, device=dev)
stem = model.blocks[0]
with torch.no_grad():
    R.FUSED_STEM = True
    y1 = stem(x).clone()
    t1 = timeit(lambda: stem(x), 20)
    R.FUSED_STEM = False
    y0 = stem(x).clone()
    t0 = timeit(lambda: stem(x), 20)
    ref = torch.nn.functional.max_pool2d(torch.relu(stem[1](stem[0](x[:32]).double().float())), 3, 2, 1)
print(f'fused stem {t1:.1f} us, MIOpen conv + tail kernel {t0:.1f} us; max |d| / max |y| fused vs unfused {float((y1 - y0).abs().max() / y0.abs().max()):.2e}; '
      f'vs torch modules (32 images) fused {float((y1[:32] - ref).abs().max() / ref.abs().max()):.2e} unfused {float((y0[:32] - ref).abs().max() / ref.abs().max()):.2e}')
flops = 2 * 256 * 64 * 112 * 112 * 147
print(f'fused: {flops / t1 / 1e6:.1f} TFLOP/s fp32-equivalent ({6 * flops / t1 / 1e6:.1f} TFLOP/s bf16 issued in six passes), {(x.numel() + y1.numel()) * 4 / t1 / 1e3:.0f} GB/s algorithmic')
from quant import _hip  # noqa: E402
w, b = R._folded_conv_bn(stem, stem[0], stem[1])
for split in (3, 2):
    t = timeit(lambda: _hip.stem_conv_pool(x, w, b, split), 20)
    print(f'split {split}: {t:.1f} us')
